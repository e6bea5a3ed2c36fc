// resample_kernels.hip -- gfx950 kernels for imageflow's resample + render path.
//
// Replaces the arithmetic behind imageflow_core::graphics::scaling::scale_and_render
// (graphics/scaling.rs:19-90): sample -> working float (graphics/color.rs:22-45), vertical then horizontal
// weighted convolution driven by PixelRowWeights tables (graphics/weights.rs:521-571,681-788), and the three
// output stages (scaling.rs:211-251 ReplaceSelf, :119-148 BlendWithMatte, :254-287 BlendWithSelf).
//
// Bound: HBM.  This is a 1-D stencil per axis, so there is no MFMA here; the design points are
//   * every source byte is read from HBM exactly once, 16 B per lane, rows fully coalesced;
//   * the vertical pass never leaves registers: each lane owns 4 source columns and a ring of K live
//     output rows; the per-row weights are wave-uniform and arrive through the scalar cache (VStep);
//   * only the 10-20x smaller vertically-reduced row goes through LDS for the horizontal pass;
//   * the sRGB->linear table lives in LDS (one ds_read per channel sample).
// Build with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, everything else rounds
// separately, exactly as the arithmetic contract in oracle/if_oracle.c (tests compare bit for bit).
#include <hip/hip_runtime.h>

#include <atomic>

#include "device.hpp"

namespace ifhip {

#ifndef IFHIP_H_UNROLL
#define IFHIP_H_UNROLL 1     // measured: 1 beats 2 and 3 (-1.7%); the chain is not latency-bound per group, code size matters
#endif
constexpr size_t kFusedLdsCap = 160 * 1024;      // gfx950: a workgroup may use the whole CU's LDS

// ------------------------------------------------------------------------------------------------------
// Output stage (shared by the fused and the generic kernels)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t uchar_clamp_ff(float v) {        // graphics/color.rs:101-108
    const double t = static_cast<double>(v) + 0.5;
    int i;
    if (t != t) i = 0;
    else if (t >= 32767.0) i = 32767;
    else if (t <= -32768.0) i = -32768;
    else i = static_cast<int>(t);
    unsigned r = static_cast<unsigned>(i) & 0xFFFFu;
    if (r > 255u) r = (v < 0.0f) ? 0u : 255u;
    return static_cast<uint8_t>(r);
}

// Tables the output stage reads: `s2f` = sRGB byte -> working float (256), `l2s` = linear -> sRGB byte (16384).
// Template parameters so that the fused kernel can hand in LDS pointers (address space known statically) and the
// generic kernels HBM pointers.
template <typename LutF, typename LutB>
struct OutTables {
    LutF s2f;
    LutB l2s;
};

// LDS-resident tables of the fused kernel.
//  * BankedLut: 32 copies of the 256-entry float table, copy b living entirely in LDS bank b
//    (dword address = idx*32 + lane%32), so the 32 lanes a ds_read_b32 services per cycle never collide,
//    whatever their indices.  A single copy costs ~3.5 LDS cycles per lane group on random pixels and made
//    the whole kernel LDS-bound (profiles/r1_v2_pmc_summary.txt).
//  * ThresholdL2S: linear->sRGB via upper_bound over the 256 thresholds of the (monotone) 16384-entry table:
//    8 dependent ds_read_u16, only ~600 times per output row, and 512 B of LDS instead of 16 KiB.
struct BankedLut {
    const float* base;      // LDS
    uint32_t lane_off;      // lane % copies
    uint32_t shift;         // log2(copies): 5 = one copy per bank; fewer copies when LDS is short (2^(5-shift)-way worst case)
    __device__ __forceinline__ float operator[](uint32_t idx) const { return base[(idx << shift) + lane_off]; }
};
struct ThresholdL2S {
    const uint16_t* thr;    // LDS, 256 entries
    const uint8_t* table;   // LDS, 16384 entries, or nullptr when LDS is short (wave-uniform choice)
    __device__ __forceinline__ uint8_t operator[](uint32_t idx) const {
        if (table) return table[idx];
        uint32_t lo = 0;                        // count of thresholds <= idx
#pragma unroll
        for (uint32_t step = 128; step > 0; step >>= 1)
            if (thr[lo + step - 1] <= idx) lo += step;
        return static_cast<uint8_t>(lo);
    }
};

template <typename LutB>
__device__ __forceinline__ uint8_t encode_channel(const ResampleArgs& a, LutB l2s, float v) {   // color.rs:61-71
    if (a.linear) {                                                                            // lut.rs:4-8
        float s = v * 16383.0f;
        s = (s != s) ? 0.0f : s;
        s = s < 0.0f ? 0.0f : s;
        s = s > 16383.0f ? 16383.0f : s;
        return l2s[static_cast<uint32_t>(s)];
    }
    return uchar_clamp_ff(255.0f * v);
}

// px: premultiplied working-space pixel (B,G,R,A).  Returns the BGRA8 word to store at the canvas pixel
// whose current content is `dst` (only read for BlendWithSelf).
template <bool ALPHA, typename LutF, typename LutB>
__device__ __forceinline__ uint32_t render_pixel(const ResampleArgs& a, float p0, float p1, float p2, float pa,
                                                 uint32_t dst, const OutTables<LutF, LutB>& tb) {
    const LutF lut = tb.s2f;
    auto encode_channel = [&](const ResampleArgs& aa, float v) -> uint32_t { return ifhip::encode_channel(aa, tb.l2s, v); };
    uint32_t b, g, r, al;
    if (!ALPHA) {
        // scaling.rs:227-232 / :267-271: alpha is not meaningful -> straight encode, alpha = 255
        b = encode_channel(a, p0); g = encode_channel(a, p1); r = encode_channel(a, p2); al = 255u;
    } else if (a.mode == IFHIP_REPLACE_SELF) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (pa > 0.0f) { c0 = p0 / pa; c1 = p1 / pa; c2 = p2 / pa; }
        b = encode_channel(a, c0); g = encode_channel(a, c1); r = encode_channel(a, c2);
        al = uchar_clamp_ff(pa * 255.0f);
    } else if (a.mode == IFHIP_BLEND_WITH_MATTE) {
        float sa = pa < 0.0f ? 0.0f : (pa > 1.0f ? 1.0f : pa);
        const float ia = (1.0f - sa) * a.matte_a;
        const float fa = ia + sa;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (fa > 0.0f) {
            c0 = (p0 + a.m0 * ia) / fa;
            c1 = (p1 + a.m1 * ia) / fa;
            c2 = (p2 + a.m2 * ia) / fa;
        }
        b = encode_channel(a, c0); g = encode_channel(a, c1); r = encode_channel(a, c2);
        al = uchar_clamp_ff(255.0f * fa);
    } else {                                                        // BlendWithSelf, scaling.rs:254-287
        if (pa > 0.994f) {
            b = encode_channel(a, p0); g = encode_channel(a, p1); r = encode_channel(a, p2); al = 255u;
        } else {
            const uint32_t da = dst >> 24;
            const float dest_coeff = (1.0f - pa) * ((1.0f / 255.0f) * static_cast<float>(static_cast<int>(da)) + 0.0f);
            const float fa = pa + dest_coeff;
            b = encode_channel(a, (p0 + dest_coeff * lut[dst & 255u]) / fa);
            g = encode_channel(a, (p1 + dest_coeff * lut[(dst >> 8) & 255u]) / fa);
            r = encode_channel(a, (p2 + dest_coeff * lut[(dst >> 16) & 255u]) / fa);
            al = uchar_clamp_ff(fa * 255.0f);
        }
    }
    return b | (g << 8) | (r << 16) | (al << 24);
}

// Canvas stores go out through inline asm on purpose.  On gfx950 loads and stores share vmcnt, and as soon as the
// compiler sees both kinds pending it treats the counter as out-of-order and drains it (s_waitcnt vmcnt(0)) at the
// next use of any loaded value -- which here would flush the D source rows every lane keeps in flight.  A store the
// compiler cannot see only makes its counted waits more conservative (vmcnt(N) with N = younger LOADS still implies
// the awaited load has returned); nothing ever reads these stores back inside the kernel, and the wave's
// outstanding stores are completed by the hardware before s_endpgm retires it.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_u32_untracked(uint32_t* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_f32x4_untracked(float4* p, float x, float y, float z, float w) {
    f32x4_t v = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <bool ALPHA, typename LutF, typename LutB>
__device__ __forceinline__ void store_pixel(const ResampleArgs& a, uint32_t img, uint32_t j, uint32_t u,
                                            float p0, float p1, float p2, float pa, const OutTables<LutF, LutB>& tb) {
    uint8_t* cp = a.canvas + static_cast<size_t>(img) * a.canvas_image_bytes
                  + static_cast<size_t>(a.y + j) * a.c_stride + static_cast<size_t>(a.x + u) * 4u;
    uint32_t* cw = reinterpret_cast<uint32_t*>(cp);           // canvas rows are 4-byte aligned (checked on host)
    uint32_t dst = 0;
    if (ALPHA && a.mode == IFHIP_BLEND_WITH_SELF) dst = *cw;
    store_u32_untracked(cw, render_pixel<ALPHA>(a, p0, p1, p2, pa, dst, tb));
    if (a.f32_dump) {
        float4* d = reinterpret_cast<float4*>(a.f32_dump) + (static_cast<size_t>(img) * a.out_h + j) * a.out_w + u;
        store_f32x4_untracked(d, p0, p1, p2, ALPHA ? pa : 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------------
// Fused kernel: one workgroup = (image, band of output rows, column strip)
//
// Vertical pass in registers, one lane = 4 source columns x K live output rows; when an output row completes, its
// vertically filtered row goes to LDS (double buffered) and the horizontal pass for it is *interleaved* into the
// following source-row steps, a few taps per step, so that its LDS latency and its strictly sequential fmaf chains
// hide under the vertical pass instead of stopping it.  One LDS-only workgroup barrier per output row.
// ------------------------------------------------------------------------------------------------------
template <int K, bool ALPHA, bool WLDS>
__global__ void __launch_bounds__(fused_max_threads(K, ALPHA ? 4 : 3))
fused_resample_kernel(const ResampleArgs a, const VStep* __restrict__ steps) {
    // `steps` is a separate __restrict__ argument (not a field of `a`) so that the compiler can prove the canvas
    // stores never clobber it and keeps the per-step 64-byte records on the scalar path.
    constexpr int C = ALPHA ? 4 : 3;
    constexpr int D = fused_shape(K, C).rows_in_flight;
    constexpr bool PIPE = fused_shape(K, C).pipelined != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t T = blockDim.x;
    uint32_t b = blockIdx.x;
    const uint32_t strip_i = b % a.n_strips; b /= a.n_strips;
    const uint32_t band = b % a.n_bands;
    const uint32_t img = b / a.n_bands;

    const Strip strip = a.strips[strip_i];
    const uint32_t n_u = strip.u1 - strip.u0;

    const FusedLds L = fused_lds_layout(n_u, strip.nquads, a.h_wu_floats, C, WLDS, a.l2s_in_lds != 0, a.lut_copies_log2);
    float* lut_banked = reinterpret_cast<float*>(smem + L.lut);      // [256][32] floats, one copy per bank
    uint16_t* thr = reinterpret_cast<uint16_t*>(smem + L.thr);       // 256 linear->sRGB thresholds
    uint4* hmeta = reinterpret_cast<uint4*>(smem + L.hmeta);         // per output column {left - cx0, taps, w offset}
    float* obuf = reinterpret_cast<float*>(smem + L.obuf);           // 2 x [n_u][4] horizontally filtered rows
    const float* hw_lds = reinterpret_cast<const float*>(smem + L.hw);
    float* inter = reinterpret_cast<float*>(smem + L.inter);         // 2 x vertically filtered row, C planes each
    const uint32_t inter_stride = L.inter_stride >> 2;               // floats per buffered row
    const uint32_t plane_pitch = L.plane_pitch;                      // floats per channel plane
    const uint32_t obuf_stride = n_u * 4u;                           // floats

    for (uint32_t i = tid; i < (256u << a.lut_copies_log2); i += T) lut_banked[i] = a.lut_in[i >> a.lut_copies_log2];
    for (uint32_t i = tid; i < 256u; i += T) thr[i] = a.l2s_thr[i];
    const uint8_t* l2s_lds = a.l2s_in_lds ? smem + L.l2s : nullptr;
    if (a.l2s_in_lds)
        for (uint32_t i = tid; i < 1024u; i += T)
            reinterpret_cast<uint4*>(smem + L.l2s)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    const BankedLut lut{lut_banked, tid & ((1u << a.lut_copies_log2) - 1u), a.lut_copies_log2};
    for (uint32_t i = tid; i < n_u; i += T) {
        uint4 m = a.h_meta[strip.u0 + i];
        m.x -= strip.cx0;
        hmeta[i] = m;
    }
    if (WLDS) {
        const float4* src4 = reinterpret_cast<const float4*>(a.h_wu);
        float4* dst4 = reinterpret_cast<float4*>(smem + L.hw);
        for (uint32_t i = tid; i < (a.h_wu_floats >> 2); i += T) dst4[i] = src4[i];
    }
    __syncthreads();
    // Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. throw away the
    // D source rows every lane keeps in flight.
    auto lds_barrier = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    const uint32_t s0 = a.band_begin[band], s1 = a.band_begin[band + 1];
    const bool lane_on = tid < strip.nquads;
    // lanes past the strip re-read its last quad instead of branching: every row load is unconditional, so the
    // number of loads in flight is known statically and the compiler can wait with vmcnt(D-1) instead of vmcnt(0)
    const uint32_t quad = lane_on ? tid : strip.nquads - 1u;
    const uint8_t* src = a.in + static_cast<size_t>(img) * a.in_image_bytes
                         + static_cast<size_t>(strip.cx0 + 4u * quad) * 4u;

    float acc[K][4][C];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[s][p][c] = 0.0f;

    auto fetch_row = [&](int y) -> uint4 {                               // y is wave-uniform; -1 = nothing needed
        const uint32_t yy = y < 0 ? 0u : static_cast<uint32_t>(y);      // (row 0 is re-read: stays in L2)
        const uint4* p = reinterpret_cast<const uint4*>(src + static_cast<size_t>(yy) * a.in_stride);
        // source frames are streamed exactly once: non-temporal loads keep them from displacing the tables in L2
        // (measured -1.7% kernel time, profiles/r1_notes.md)
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    };

    // sample -> working float for the 4 pixels of one 16-byte load (arithmetic contract step 1)
    auto convert = [&](const uint4& q, float (&v)[4][C]) {
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t px = w4[p];
            v[p][0] = lut[px & 255u];
            v[p][1] = lut[(px >> 8) & 255u];
            v[p][2] = lut[(px >> 16) & 255u];
            if (ALPHA) {
                const float af = static_cast<float>(px >> 24) * (1.0f / 255.0f);
                v[p][0] = v[p][0] * af;
                v[p][1] = v[p][1] * af;
                v[p][2] = v[p][2] * af;
                v[p][C - 1] = af;
            }
        }
    };

    // ---- horizontal pass of one output row: chain idx = (output column ul, channel c), the strictly ascending
    // fmaf sum over its taps (arithmetic contract step 3).  Runs right after the row hand-over barrier on the
    // lowest lanes.  Samples come from the channel's plane, weights from the output's (de-duplicated) row, both as
    // aligned 16-byte LDS reads of 4 taps; the first group may begin with +0 weights (columns before the first tap),
    // the last group is predicated on the number of valid taps.
    const uint32_t n_chain = n_u * C;
    auto h_run_row = [&](const float* vrow, float* orow) {
        for (uint32_t idx = tid; idx < n_chain; idx += T) {
            const uint32_t ul = idx / C, c = idx - ul * C;
            const uint4 m = hmeta[ul];
            const float4* sp = reinterpret_cast<const float4*>(vrow + c * plane_pitch + m.x);
            const float4* wp = reinterpret_cast<const float4*>((WLDS ? hw_lds : a.h_wu) + m.z);
            const uint32_t last = m.y - 1u;
            float h = 0.0f;
#pragma unroll IFHIP_H_UNROLL
            for (uint32_t q = 0; q < last; ++q) {
                const float4 w = wp[q];
                const float4 x = sp[q];
                h = __builtin_fmaf(w.x, x.x, h);
                h = __builtin_fmaf(w.y, x.y, h);
                h = __builtin_fmaf(w.z, x.z, h);
                h = __builtin_fmaf(w.w, x.w, h);
            }
            {
                const float4 w = wp[last];
                const float4 x = sp[last];
                h = __builtin_fmaf(w.x, x.x, h);
                if (m.w > 1u) h = __builtin_fmaf(w.y, x.y, h);
                if (m.w > 2u) h = __builtin_fmaf(w.z, x.z, h);
                if (m.w > 3u) h = __builtin_fmaf(w.w, x.w, h);
            }
            orow[ul * 4u + c] = h;
        }
    };
    int h_out_row = -1;              // output row whose horizontal result is waiting in obuf (uniform), -1: none
    auto h_store_row = [&](uint32_t j, const float* orow) {      // output stage of a horizontally filtered row
        // the lanes at the top of the workgroup take it: the chains sit on the lowest lanes
        for (uint32_t ul = T - 1u - tid; ul < n_u; ul += T) {
            const float4 o = *reinterpret_cast<const float4*>(orow + ul * 4u);
            const OutTables<BankedLut, ThresholdL2S> tb{lut, ThresholdL2S{thr, l2s_lds}};
            store_pixel<ALPHA>(a, img, j, strip.u0 + ul, o.x, o.y, o.z, ALPHA ? o.w : 1.0f, tb);
        }
    };

    // Software pipeline over steps (one step = one source row):
    //   raw[D]  : D source rows in flight per lane (16 B each), refilled in place -> fixed registers, vmcnt(D-1)
    //   vbuf[2] : (PIPE) converted floats of the current / the next step; the 12-16 LUT reads of step i+1 are issued
    //             before the FMAs of step i, so their LDS latency hides under the lane's own arithmetic
    //   rec[2]  : (PIPE) the 64-byte step records of the current / the next step (scalar loads, same overlap)
    // Register-heavier rings use the plain form (!PIPE): convert, refill, accumulate, one step at a time.
    // The host pads every band to a multiple of D steps, so the unrolled group has no early exit and every buffer
    // index below is a compile-time constant.
    uint4 raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        raw[d] = fetch_row(steps[s0 + d].y);
        __builtin_amdgcn_sched_barrier(0);      // keep issue order raw[0..D-1]: the loop's vmcnt(D-1) relies on it
    }
    float vbuf[PIPE ? 2 : 1][4][C];
    VStep rec[PIPE ? 2 : 1];
    if (PIPE) {
        rec[0] = steps[s0];
        convert(raw[0], vbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
        raw[0] = fetch_row((s0 + D < s1) ? steps[s0 + D].y : -1);
        __builtin_amdgcn_sched_barrier(0);
    }

    for (uint32_t sb = s0; sb < s1; sb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int cur = PIPE ? (d & 1) : 0, nxt = PIPE ? (cur ^ 1) : 0;
            const uint32_t si = sb + d;
            if (PIPE) {
                // ---- stage A: start step si+1 (record, LUT gathers), refill its row slot for step si+1+D ----
                const int slot_next = (d + 1) % D;
                rec[nxt] = steps[(si + 1 < s1) ? si + 1 : si];
                convert(raw[slot_next], vbuf[nxt]);
                __builtin_amdgcn_sched_barrier(0);
                raw[slot_next] = fetch_row(rec[cur].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                rec[0] = steps[si];
                convert(raw[d], vbuf[0]);
                // The bytes of raw[d] are consumed; only now re-issue the load into the same registers (row of step
                // si+D).  Issuing it earlier would overlap the two live ranges and make the compiler rotate the
                // registers with copies (and a vmcnt(0) drain) at the loop back edge.
                __builtin_amdgcn_sched_barrier(0);
                raw[d] = fetch_row(rec[0].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stage B: finish step si ----
            const VStep& st = rec[cur];
            float (&v)[4][C] = vbuf[cur];
#if defined(IFHIP_EXP_LOAD_ONLY)   // experiment: stream rows, no arithmetic (NOT a product path)
            acc[0][0][0] += v[0][0] + v[1][1] + v[2][2] + v[3][0];
#else
            // Every ring slot accumulates unconditionally: a slot outside its window holds exactly +0.0f (initial
            // value / reset at flush) and has weight +0.0f in the step record, and fmaf(+0, v, +0) == +0 for the
            // finite non-negative v we feed it, so the result is bit-identical to skipping the slot -- without
            // K scalar branches (and their instruction-fetch bubbles) per source row.
            if (st.y >= 0) {
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const float w = st.w[s];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[s][p][c] = __builtin_fmaf(w, v[p][c], acc[s][p][c]);
                }
            }
#endif
#if defined(IFHIP_EXP_NO_H)        // experiment: vertical pass only (NOT a product path)
            if (st.flush_slot >= 0 && st.out_row == 0x7fffffff) {
#else
            if (st.flush_slot >= 0) {
#endif
                // ---- output row j's vertical pass is complete: hand its row to the horizontal pass ----
                const uint32_t j = static_cast<uint32_t>(st.out_row);
                float* dst_row = inter + (j & 1u) * inter_stride;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st.flush_slot == s) {
                        if (lane_on) {
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                *reinterpret_cast<float4*>(dst_row + c * plane_pitch + 4u * tid) =
                                    make_float4(acc[s][0][c], acc[s][1][c], acc[s][2][c], acc[s][3][c]);
                        }
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[s][p][c] = 0.0f;
                    }
                }
                // One barrier per output row.  After it: row j's vertical result (inter[j&1]) and row j-1's horizontal
                // result (obuf[(j-1)&1]) are complete.  inter[j&1] is next written at row j+2 and obuf[(j-1)&1] by
                // row j+1's chains, both only after every wave has passed the barrier of row j+1, i.e. after every
                // wave has finished reading them.
                lds_barrier();
#if !defined(IFHIP_EXP_NO_STORE)
                if (h_out_row >= 0) h_store_row(static_cast<uint32_t>(h_out_row), obuf + (static_cast<uint32_t>(h_out_row) & 1u) * obuf_stride);
#endif
                h_out_row = static_cast<int>(j);
#if !defined(IFHIP_EXP_NO_CHAIN)
                h_run_row(dst_row, obuf + (j & 1u) * obuf_stride);
#endif
            }
        }
    }
    // drain: the last output row of the band still has its horizontal pass and output stage to do
    if (h_out_row >= 0) {
        lds_barrier();
        h_store_row(static_cast<uint32_t>(h_out_row), obuf + (static_cast<uint32_t>(h_out_row) & 1u) * obuf_stride);
    }
}

// ------------------------------------------------------------------------------------------------------
// Generic two-pass kernels (any ratio, any alignment): vertical gather into an HBM f32 scratch, then
// horizontal gather + output stage.  Same arithmetic contract, used when the fused kernel's preconditions
// (<= kMaxSlots live rows, 16-byte aligned rows) do not hold, and as the fused kernel's on-device cross-check.
// ------------------------------------------------------------------------------------------------------
template <bool ALPHA>
__global__ void __launch_bounds__(256)
vpass_generic_kernel(const ResampleArgs a, float4* scratch, uint32_t img0) {
    const uint32_t xcol = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    const uint32_t img = img0 + blockIdx.z;
    if (xcol >= a.in_w) return;
    const uint8_t* col = a.in + static_cast<size_t>(img) * a.in_image_bytes + static_cast<size_t>(xcol) * 4u;
    const uint32_t left = a.v_left[j], n = a.v_count[j];
    const float* w = a.v_w + a.v_off[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t* p = col + static_cast<size_t>(left + k) * a.in_stride;
        float f0 = a.lut_in[p[0]], f1 = a.lut_in[p[1]], f2 = a.lut_in[p[2]], f3 = 1.0f;
        if (ALPHA) {
            f3 = static_cast<float>(p[3]) * (1.0f / 255.0f);
            f0 = f0 * f3; f1 = f1 * f3; f2 = f2 * f3;
        }
        const float wk = w[k];
        s0 = __builtin_fmaf(wk, f0, s0);
        s1 = __builtin_fmaf(wk, f1, s1);
        s2 = __builtin_fmaf(wk, f2, s2);
        if (ALPHA) s3 = __builtin_fmaf(wk, f3, s3);
    }
    scratch[(static_cast<size_t>(blockIdx.z) * a.out_h + j) * a.in_w + xcol] = make_float4(s0, s1, s2, ALPHA ? s3 : 1.0f);
}

template <bool ALPHA>
__global__ void __launch_bounds__(256)
hpass_generic_kernel(const ResampleArgs a, const float4* scratch, uint32_t img0) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    const uint32_t img = img0 + blockIdx.z;
    if (u >= a.out_w) return;
    const float4* row = scratch + (static_cast<size_t>(blockIdx.z) * a.out_h + j) * a.in_w;
    const uint32_t left = a.h_left[u], n = a.h_count[u];
    const float* w = a.h_w + a.h_off[u];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (uint32_t k = 0; k < n; ++k) {
        const float4 v = row[left + k];
        const float wk = w[k];
        s0 = __builtin_fmaf(wk, v.x, s0);
        s1 = __builtin_fmaf(wk, v.y, s1);
        s2 = __builtin_fmaf(wk, v.z, s2);
        if (ALPHA) s3 = __builtin_fmaf(wk, v.w, s3);
    }
    const OutTables<const float*, const uint8_t*> tb{a.lut_in, a.l2s};
    store_pixel<ALPHA>(a, img, j, u, s0, s1, s2, ALPHA ? s3 : 1.0f, tb);
}

// ------------------------------------------------------------------------------------------------------
// Flatten: graphics/blend.rs:6-59, one lane per pixel, 4 B in / 4 B out, in place.
// ------------------------------------------------------------------------------------------------------
struct MatteArgs {
    uint8_t* bgra;
    size_t image_bytes;
    uint32_t w, h, stride, n_images;
    uint32_t matte;             // B,G,R,A bytes
    float mb, mg, mr, ma;       // linear matte colour, matte alpha / 255
    const float* s2l;
    const uint8_t* l2s;
};

__global__ void __launch_bounds__(256) apply_matte_kernel(const MatteArgs a) {
    __shared__ float lut[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) lut[i] = a.s2l[i];
    __syncthreads();
    const uint32_t xx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t yy = blockIdx.y;
    const uint32_t img = blockIdx.z;
    if (xx >= a.w) return;
    uint32_t* p = reinterpret_cast<uint32_t*>(a.bgra + static_cast<size_t>(img) * a.image_bytes
                                              + static_cast<size_t>(yy) * a.stride) + xx;
    const uint32_t px = *p;
    const uint32_t pa = px >> 24;
    if (pa == 255u) return;
    if (pa == 0u) { *p = a.matte; return; }
    const float paf = static_cast<float>(static_cast<int>(pa)) * (1.0f / 255.0f);
    const float ma = (1.0f - paf) * a.ma;
    const float fa = ma + paf;
    auto enc = [&](float v) -> uint32_t {
        float s = v * 16383.0f;
        s = (s != s) ? 0.0f : s;
        s = s < 0.0f ? 0.0f : s;
        s = s > 16383.0f ? 16383.0f : s;
        return a.l2s[static_cast<uint32_t>(s)];
    };
    const uint32_t nb = enc((lut[px & 255u] * paf + a.mb * ma) / fa);
    const uint32_t ng = enc((lut[(px >> 8) & 255u] * paf + a.mg * ma) / fa);
    const uint32_t nr = enc((lut[(px >> 16) & 255u] * paf + a.mr * ma) / fa);
    const uint32_t na = uchar_clamp_ff(255.0f * fa);
    *p = nb | (ng << 8) | (nr << 16) | (na << 24);
}

// ------------------------------------------------------------------------------------------------------
// Launchers (called from api.cpp)
// ------------------------------------------------------------------------------------------------------
template <int K, bool ALPHA, bool WLDS>
static hipError_t launch_fused_kaw(const ResampleArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    // raise the dynamic-LDS cap once per kernel variant and device (it is sticky), not on every launch
    static std::atomic<size_t> cap[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>& c = cap[dev & 15];
    if (c.load(std::memory_order_relaxed) < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_resample_kernel<K, ALPHA, WLDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsCap));
        c.store(kFusedLdsCap, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((fused_resample_kernel<K, ALPHA, WLDS>), grid, block, lds, st, a, a.steps);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_fused_k(const ResampleArgs& a, bool alpha, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    const bool wl = a.h_w_in_lds != 0;
    if (alpha) return wl ? launch_fused_kaw<K, true, true>(a, grid, block, lds, st)
                         : launch_fused_kaw<K, true, false>(a, grid, block, lds, st);
    return wl ? launch_fused_kaw<K, false, true>(a, grid, block, lds, st)
              : launch_fused_kaw<K, false, false>(a, grid, block, lds, st);
}

hipError_t launch_fused(const ResampleArgs& a, int slots, bool alpha, uint32_t grid, uint32_t block, size_t lds,
                        hipStream_t st) {
    const dim3 g(grid), b(block);
    switch (slots) {            // K must equal the ring size exactly: every slot accumulates on every row
    case 1: return launch_fused_k<1>(a, alpha, g, b, lds, st);
    case 2: return launch_fused_k<2>(a, alpha, g, b, lds, st);
    case 3: return launch_fused_k<3>(a, alpha, g, b, lds, st);
    case 4: return launch_fused_k<4>(a, alpha, g, b, lds, st);
    case 5: return launch_fused_k<5>(a, alpha, g, b, lds, st);
    case 6: return launch_fused_k<6>(a, alpha, g, b, lds, st);
    case 7: return launch_fused_k<7>(a, alpha, g, b, lds, st);
    case 8: return launch_fused_k<8>(a, alpha, g, b, lds, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_generic(const ResampleArgs& a, bool alpha, float4* scratch, uint32_t img0, uint32_t n_img,
                          hipStream_t st) {
    const dim3 bv(256), gv((a.in_w + 255u) / 256u, a.out_h, n_img);
    const dim3 bh(256), gh((a.out_w + 255u) / 256u, a.out_h, n_img);
    if (alpha) {
        hipLaunchKernelGGL((vpass_generic_kernel<true>), gv, bv, 0, st, a, scratch, img0);
        hipLaunchKernelGGL((hpass_generic_kernel<true>), gh, bh, 0, st, a, scratch, img0);
    } else {
        hipLaunchKernelGGL((vpass_generic_kernel<false>), gv, bv, 0, st, a, scratch, img0);
        hipLaunchKernelGGL((hpass_generic_kernel<false>), gh, bh, 0, st, a, scratch, img0);
    }
    return hipGetLastError();
}

hipError_t launch_apply_matte(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                              uint32_t stride, uint32_t matte, float mb, float mg, float mr, float ma,
                              const float* s2l, const uint8_t* l2s, hipStream_t st) {
    MatteArgs m{d_bgra, image_bytes, w, h, stride, n_images, matte, mb, mg, mr, ma, s2l, l2s};
    const dim3 block(256), grid((w + 255u) / 256u, h, n_images);
    hipLaunchKernelGGL(apply_matte_kernel, grid, block, 0, st, m);
    return hipGetLastError();
}

}  // namespace ifhip
