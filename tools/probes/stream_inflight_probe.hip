// stream_inflight_probe.hip -- how fast a CU of gfx950 streams from HBM as a function of the bytes it keeps in flight, and
// what a compute pause between bursts of loads costs.  The question behind it: the fused
// resampler's workgroups alternate between streaming source rows (D rows of 16 B per lane in flight) and a pixel loop that
// issues no loads; rows in flight 4 / 6 / 8 made no difference there.
// One workgroup of 1024 lanes per CU (the LDS request makes sure of that), each lane streams its own 16-byte column of
// consecutive "rows" (row pitch = 16 KiB: a row is one fully coalesced 16 KiB line of the workgroup, like a 4K BGRA row)
// with D loads in flight (refilled in place, vmcnt(D-1) waits), XORs what it reads, and every `period` rows spins for
// `pause` iterations of dependent VALU work (no memory instruction), like the pixel loop.
// Printed per configuration: GB/s over the whole chip (grid = 256 workgroups) and for a handful of workgroups alone.
// NOT part of the product.  Build: hipcc --offload-arch=gfx950 -O2 -o stream_inflight_probe stream_inflight_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ void __launch_bounds__(1024) stream_kernel(const u32x4* __restrict__ src, size_t rows_per_wg, uint32_t* sink, int period, int pause,
                                                      u32x4* __restrict__ dst, int wperiod) {
    extern __shared__ unsigned char lds[];                       // (requested size keeps it at one workgroup per CU)
    typedef __attribute__((address_space(1))) const u32x4 gvec;
    const size_t pitch = 1024;                                   // u32x4 per row of the workgroup
    gvec* base = reinterpret_cast<gvec*>(reinterpret_cast<uintptr_t>(src + static_cast<size_t>(blockIdx.x) * rows_per_wg * pitch + threadIdx.x));
    u32x4 raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        raw[d] = __builtin_nontemporal_load(base + static_cast<size_t>(d) * pitch);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t acc = threadIdx.x;
    float spin = 1.0f + threadIdx.x;
    int since = 0, wsince = 0;
    size_t wrow = 0;
    typedef __attribute__((address_space(1))) u32x4 gout;
    gout* obase = reinterpret_cast<gout*>(reinterpret_cast<uintptr_t>(dst + static_cast<size_t>(blockIdx.x) * rows_per_wg * pitch + threadIdx.x));
    for (size_t r = 0; r + D <= rows_per_wg; r += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 v = raw[d];                              // (the compiler waits with vmcnt(D-1): loads return in order)
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            __builtin_amdgcn_sched_barrier(0);
            const size_t next = r + d + D;
            raw[d] = __builtin_nontemporal_load(base + (next < rows_per_wg ? next : rows_per_wg - 1) * pitch);
            __builtin_amdgcn_sched_barrier(0);
            if (wperiod > 0 && ++wsince == wperiod) {            // one 16 KiB row of output per `wperiod` rows read (the canvas stores)
                wsince = 0;
                // (through inline asm, as the product's canvas stores: loads and stores share vmcnt on gfx950, and a store the
                // compiler sees makes it drain the counter -- vmcnt(0) -- at the next use of a loaded value)
                const u32x4 ov = {acc, acc, acc, acc};
                gout* op = obase + wrow * pitch;
#ifdef STORE4                                                     // the same 16 KiB as four 4-byte stores per lane (the product's canvas stores are 4 bytes per lane)
                uint32_t* o4 = reinterpret_cast<uint32_t*>(dst + static_cast<size_t>(blockIdx.x) * rows_per_wg * pitch + wrow * pitch) + threadIdx.x;
                asm volatile("global_store_dword %0, %1, off nt\n\tglobal_store_dword %2, %1, off nt\n\tglobal_store_dword %3, %1, off nt\n\tglobal_store_dword %4, %1, off nt"
                             : : "v"(o4), "v"(acc), "v"(o4 + 1024), "v"(o4 + 2048), "v"(o4 + 3072) : "memory");
#else
                asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(op), "v"(ov) : "memory");
#endif
                ++wrow;
            }
            if (pause > 0 && ++since == period) {                // wave-uniform: the "pixel loop" -- no loads are issued in here
                since = 0;
                for (int i = 0; i < pause; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(spin));
            }
        }
    }
    if (acc == 0x12345678u && spin == 3.0f) sink[0] = acc;       // (never true: keeps the work alive)
    if (lds[threadIdx.x & 15] == 77 && acc == 1u) sink[1] = 1u;
}

template <int D>
static double run(const u32x4* d_src, size_t rows_per_wg, int grid, uint32_t* d_sink, int period, int pause, double* ms_out, u32x4* d_dst = nullptr, int wperiod = 0) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(stream_kernel<D>, dim3(grid), dim3(1024), 100 * 1024, 0, d_src, rows_per_wg, d_sink, period, pause, d_dst, wperiod);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(stream_kernel<D>, dim3(grid), dim3(1024), 100 * 1024, 0, d_src, rows_per_wg, d_sink, period, pause, d_dst, wperiod);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (ms_out) *ms_out = ms;
    return static_cast<double>(grid) * rows_per_wg * 16384.0 / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t rows_per_wg = 2048;                             // 32 MiB per workgroup: one 4K BGRA frame
    const int full = 256;
    u32x4* d_src = nullptr;
    uint32_t* d_sink = nullptr;
    if (hipMalloc(&d_src, static_cast<size_t>(full) * rows_per_wg * 16384) != hipSuccess || hipMalloc(&d_sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d_src, 1, static_cast<size_t>(full) * rows_per_wg * 16384);
    printf("# one workgroup of 1024 lanes per CU, 16 B per lane and row, rows of 16 KiB, 2048 rows per workgroup (a 4K BGRA frame)\n");
    printf("# D = loads in flight per lane (bytes in flight per CU = D x 16 KiB); pause = dependent v_fma per lane every `period` rows\n");
    printf("%-6s %-8s %-8s %14s %14s %18s\n", "D", "period", "pause", "256 WGs GB/s", "ms", "16 WGs GB/s per WG");
    const int pauses[][2] = {{0, 0}, {2, 80}};
    for (auto& pp : pauses) {
        double ms = 0;
#define ROW(Dv) { const double all = run<Dv>(d_src, rows_per_wg, full, d_sink, pp[0], pp[1], &ms); const double few = run<Dv>(d_src, rows_per_wg, 16, d_sink, pp[0], pp[1], nullptr); \
                  printf("%-6d %-8d %-8d %14.0f %14.3f %18.1f\n", Dv, pp[0], pp[1], all, ms, few / 16.0); }
        ROW(1) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12) ROW(16)
#undef ROW
    }
    // reads and writes together: one row written per `wperiod` rows read (D = 6, no pause); GB/s counts both directions
    u32x4* d_dst = nullptr;
    if (hipMalloc(&d_dst, static_cast<size_t>(full) * rows_per_wg * 16384) == hipSuccess) {
        printf("# reads + writes: a 16 KiB row written per wperiod rows read, D = 6, no pause; GB/s of reads + writes, 256 workgroups\n");
        for (int wp : {0, 16, 8, 6, 4, 3, 2, 1}) {
            double ms = 0;
            run<6>(d_src, rows_per_wg, full, d_sink, 0, 0, &ms, d_dst, wp);
            const double bytes = static_cast<double>(full) * rows_per_wg * 16384.0 * (1.0 + (wp ? 1.0 / wp : 0.0));
            printf("wperiod %-3d write share %5.1f %%  %8.3f ms  %8.0f GB/s\n", wp, wp ? 100.0 / (wp + 1.0) : 0.0, ms, bytes / (ms * 1e-3) / 1e9);
        }
        hipFree(d_dst);
    }
    hipFree(d_src); hipFree(d_sink);
    return 0;
}
