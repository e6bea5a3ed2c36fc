// issue_rate_probe.hip -- what one instruction of the resampler's inner loops costs on gfx950, measured: a wave runs
// a long stream of ONE instruction kind on 16 independent register chains, bracketed by s_memtime; 1, 2 and 4 waves per
// SIMD.  Printed: shader cycles per instruction as one wave sees them, and per SIMD (wave cycles / waves per SIMD).
// Also: MFMA and VALU mixed in one wave, and an MFMA wave beside a VALU wave on the same SIMD.
// NOT part of the product.  Build: hipcc --offload-arch=gfx950 -O2 -o issue_rate_probe issue_rate_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Mode { CND64, CMP_CND, ADD_U32, LSHL_ADD, AND_B32, BFE, ALIGNBIT, MAD_U24, MUL_I24, MUL_LO, MAD_U64, ADD3, PERM, SDWA_AND, DS_READ_B128, DS_WRITE_B32, FMA, PK_FMA, PK_FMA_SGPR, DOT4, CVT_UB3, MUL, PK_MUL, CNDMASK, MOV, MFMA, MFMA_PKFMA, MFMA_FMA, MFMA_2FMA, DS_READ, SPLIT_MFMA_PKFMA,
            SPLIT_PKFMA_PKFMA, SPLIT_MFMA_MFMA, N_MODES };
static const char* kNames[N_MODES] = {"v_cndmask_b32_e64 (sgpr mask)", "v_cmp_lt_u32 + v_cndmask (vcc)", "v_add_u32", "v_lshl_add_u32", "v_and_b32", "v_bfe_u32", "v_alignbit_b32", "v_mad_u32_u24", "v_mul_i32_i24", "v_mul_lo_u32", "v_mad_u64_u32", "v_add3_u32", "v_perm_b32", "v_and_b32 sdwa byte", "ds_read_b128 (16 in flight)", "ds_write_b32", "v_fma_f32", "v_pk_fma_f32 (vgpr)", "v_pk_fma_f32 (sgpr weight)", "v_dot4_u32_u8", "v_cvt_f32_ubyte3", "v_mul_f32", "v_pk_mul_f32",
                                      "v_cndmask_b32", "v_mov_b32", "v_mfma_f32_4x4x1", "mfma + pk_fma alternating", "mfma + fma alternating",
                                      "mfma + 2 fma", "ds_read_b32 (16 in flight)", "waves 0-3 mfma | waves 4-7 pk_fma ",
                                      "waves 0-3 pk_fma | waves 4-7 pk_fma", "waves 0-3 mfma | waves 4-7 mfma"};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ void __launch_bounds__(1024) probe(uint64_t* cycles, float* sink, int iters, float wv, uint32_t seed) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = static_cast<float>(i);
    __syncthreads();
    f32x2 a[16];
    f32x4 m[16];
    float s[16];
    uint32_t u[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = f32x2{0.5f + i, 0.25f + i};
        m[i] = f32x4{1.0f * i, 2.0f, 3.0f, 4.0f};
        s[i] = 0.125f * i + threadIdx.x;
        u[i] = seed * (i + 1) + threadIdx.x;
    }
    const f32x2 bv = {1.0001f, 0.9999f};
    const float b1 = 1.0001f;
    const uint64_t wpair = __builtin_amdgcn_readfirstlane(__float_as_uint(wv));     // low dword of an SGPR pair, broadcast by op_sel_hi
    const uint32_t addr = (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6) * 256u;
    const uint32_t addr4 = (threadIdx.x & 63u) * 16u;
    const uint64_t mask64 = __builtin_amdgcn_readfirstlane(seed) * 0x100000001ull;
    uint64_t w64[16];
    for (int i = 0; i < 16; ++i) w64[i] = seed + i;
    const int wave = threadIdx.x >> 6;
    int mode = MODE;
    if (MODE == SPLIT_MFMA_PKFMA) mode = wave < 4 ? MFMA : PK_FMA;
    if (MODE == SPLIT_PKFMA_PKFMA) mode = PK_FMA;
    if (MODE == SPLIT_MFMA_MFMA) mode = MFMA;
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        if (mode == CND64) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b1), "s"(mask64));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == CMP_CND) {
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(u[i]) : "v"(seed), "v"(addr) : "vcc");
            REP16(X) REP16(X)
#undef X
        } else if (mode == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == LSHL_ADD) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == AND_B32) {
#define X(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == BFE) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == ALIGNBIT) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(seed), "v"(addr));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MAD_U24) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(seed), "v"(addr));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MUL_I24) {
#define X(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MUL_LO) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MAD_U64) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w64[i]) : "v"(seed), "v"(addr) : "vcc");
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == ADD3) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(seed), "v"(addr));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(seed), "v"(addr));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == SDWA_AND) {
#define X(i) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(u[i]) : "v"(seed));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == DS_READ_B128) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m[i]) : "v"(addr4), "n"((i) * 1024));
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
#undef X
        } else if (mode == DS_WRITE_B32) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(s[i]), "n"((i) * 1024) : "memory");
            REP16(X) REP16(X) REP16(X) REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
#undef X
        } else if (mode == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(b1), "v"(wv));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bv), "v"(bv));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == PK_FMA_SGPR) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(wpair), "v"(bv));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == DOT4) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(seed), "s"(0x80u));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == CVT_UB3) {
#define X(i) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(s[i]) : "v"(u[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s[i]) : "v"(b1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(bv));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b1) : );
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(b1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MFMA) {
#define X(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m[i]) : "v"(wv), "v"(b1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MFMA_PKFMA) {
#define X(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %4, %4, %1" : "+v"(m[i]), "+v"(a[i]) : "v"(wv), "v"(b1), "v"(bv));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MFMA_FMA) {
#define X(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1" : "+v"(m[i]), "+v"(s[i]) : "v"(wv), "v"(b1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == MFMA_2FMA) {
#define X(i) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %3, %4, %0\n v_fma_f32 %1, %4, %3, %1\n v_fma_f32 %2, %4, %3, %2" : "+v"(m[i]), "+v"(s[i]), "+v"(a[i].x) : "v"(wv), "v"(b1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if (mode == DS_READ) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[i]) : "v"(addr), "n"((i) * 1024));
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
            REP16(X)
            asm volatile("s_waitcnt lgkmcnt(0)");
#undef X
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += static_cast<float>(w64[i]) + a[i].x + a[i].y + m[i].x + m[i].y + m[i].z + m[i].w + s[i] + static_cast<float>(u[i]);
    if (acc == 12345.678f) sink[0] = acc + lds[threadIdx.x & 4095];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE>
static void run(int block, uint64_t* d_cycles, float* d_sink) {
    const int grid = 256, iters = 256;
    const int waves = block / 64;
    std::vector<uint64_t> h(grid * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<grid, block>>>(d_cycles, d_sink, iters, 0.5f, 77u);
    hipEventRecord(e0);
    probe<MODE><<<grid, block>>>(d_cycles, d_sink, iters, 0.5f, 77u);
    hipEventRecord(e1);
    hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    const int per_iter = (MODE == MFMA_2FMA) ? 64 * 3 : (MODE == MFMA_PKFMA || MODE == MFMA_FMA) ? 64 * 2 : 64;   // (CMP_CND: 32 pairs = 64)
    double lo[2] = {0, 0};
    int n[2] = {0, 0};
    for (size_t i = 0; i < h.size(); ++i) {
        const int half = (MODE >= SPLIT_MFMA_PKFMA && (i % waves) >= 4) ? 1 : 0;
        lo[half] += static_cast<double>(h[i]); ++n[half];
    }
    const double wps = waves / 4.0;
    for (int half = 0; half < 2; ++half) {
        if (!n[half]) continue;
        const double cyc = lo[half] / n[half] / (static_cast<double>(iters) * per_iter);
        printf("%-48s %4d lanes (%g waves/SIMD)%s: %7.2f cycles per instruction for a wave, %6.2f per SIMD   (kernel %.3f ms, %.0f MHz if s_memtime ticks are shader cycles)\n",
               kNames[MODE], block, wps, MODE >= SPLIT_MFMA_PKFMA ? (half ? " waves 4-7" : " waves 0-3") : "", cyc, cyc / wps, ms,
               lo[half] / n[half] / (ms * 1e3));
    }
}

template <int MODE>
static void sweep(uint64_t* d_cycles, float* d_sink) {
    if (MODE >= SPLIT_MFMA_PKFMA) { run<MODE>(512, d_cycles, d_sink); return; }
    for (int block : {256, 512, 1024}) run<MODE>(block, d_cycles, d_sink);
}

int main() {
    uint64_t* d_cycles;
    float* d_sink;
    hipMalloc(&d_cycles, 256 * 16 * 8);
    hipMalloc(&d_sink, 64);
    sweep<CND64>(d_cycles, d_sink); sweep<CMP_CND>(d_cycles, d_sink); sweep<ADD_U32>(d_cycles, d_sink); sweep<LSHL_ADD>(d_cycles, d_sink);
    sweep<AND_B32>(d_cycles, d_sink); sweep<BFE>(d_cycles, d_sink); sweep<ALIGNBIT>(d_cycles, d_sink); sweep<MAD_U24>(d_cycles, d_sink);
    sweep<MUL_I24>(d_cycles, d_sink); sweep<MUL_LO>(d_cycles, d_sink); sweep<MAD_U64>(d_cycles, d_sink); sweep<ADD3>(d_cycles, d_sink);
    sweep<PERM>(d_cycles, d_sink); sweep<SDWA_AND>(d_cycles, d_sink); sweep<DS_READ_B128>(d_cycles, d_sink); sweep<DS_WRITE_B32>(d_cycles, d_sink);
    sweep<FMA>(d_cycles, d_sink);
    sweep<PK_FMA>(d_cycles, d_sink);
    sweep<PK_FMA_SGPR>(d_cycles, d_sink);
    sweep<DOT4>(d_cycles, d_sink);
    sweep<CVT_UB3>(d_cycles, d_sink);
    sweep<MUL>(d_cycles, d_sink);
    sweep<PK_MUL>(d_cycles, d_sink);
    sweep<CNDMASK>(d_cycles, d_sink);
    sweep<MOV>(d_cycles, d_sink);
    sweep<MFMA>(d_cycles, d_sink);
    sweep<MFMA_PKFMA>(d_cycles, d_sink);
    sweep<MFMA_FMA>(d_cycles, d_sink);
    sweep<MFMA_2FMA>(d_cycles, d_sink);
    sweep<DS_READ>(d_cycles, d_sink);
    sweep<SPLIT_MFMA_PKFMA>(d_cycles, d_sink);
    sweep<SPLIT_PKFMA_PKFMA>(d_cycles, d_sink);
    sweep<SPLIT_MFMA_MFMA>(d_cycles, d_sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    return 0;
}
