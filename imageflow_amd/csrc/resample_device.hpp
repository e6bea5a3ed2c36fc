// resample_device.hpp -- device-side pieces shared by the fused kernel (resample_fused.hip, one translation unit per
// ring size K) and the generic kernels (resample_kernels.hip): the output stage of scale_and_render and the LDS tables.
#pragma once
#include <hip/hip_runtime.h>

#include "device.hpp"

namespace ifhip {

constexpr size_t kFusedLdsCap = 160 * 1024;      // gfx950: a workgroup may use the whole CU's LDS

// ------------------------------------------------------------------------------------------------------
// Output stage (shared by the fused and the generic kernels)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t uchar_clamp_ff(float v) {        // graphics/color.rs:101-108
    // `(v as f64 + 0.5) as i16 as u16`, > 255 -> (v < 0 ? 0 : 255), restated without f64: negative inputs and NaN give
    // 0, everything else min(255, floor(v) + (frac(v) >= 0.5)) -- floor and the fraction are exact in f32.  Equal to the
    // f64 form for all 2^32 float bit patterns (exhaustive host check, tests/test_oracle_color.py samples it).
    const float c = __builtin_fminf(v, 300.0f);
    const float f = __builtin_floorf(c);
    int i = static_cast<int>(f) + ((c - f) >= 0.5f ? 1 : 0);
    i = i > 255 ? 255 : i;
    return (v >= 0.0f) ? static_cast<uint8_t>(i) : static_cast<uint8_t>(0);
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

// the whole 16 KiB table, known to be there (fused kernel, fast pass: asked once per output row, not per channel)
struct DirectL2S {
    const uint8_t* table;   // LDS, 16384 entries
    __device__ __forceinline__ uint8_t operator[](uint32_t idx) const { return table[idx]; }
};

// LIN: 1 = the working space is known to be linear at compile time, -1 = ask the argument block
template <int LIN = -1, typename LutB>
__device__ __forceinline__ uint8_t encode_channel(const ResampleArgs& a, LutB l2s, float v) {   // color.rs:61-71
    if (LIN == 1 || (LIN < 0 && a.linear)) {                                                   // lut.rs:4-8
        // (v * 16383).clamp(0, 16383) as usize, NaN -> 0: max(NaN, 0) = 0 (maxNum), two instructions instead of three
        // compare/select pairs
        const float s = __builtin_fminf(__builtin_fmaxf(v * 16383.0f, 0.0f), 16383.0f);
        return l2s[static_cast<uint32_t>(s)];
    }
    return uchar_clamp_ff(255.0f * v);
}

// px: premultiplied working-space pixel (B,G,R,A).  Returns the BGRA8 word to store at the canvas pixel
// whose current content is `dst` (only read for BlendWithSelf).
template <bool ALPHA, int LIN = -1, typename LutF, typename LutB>
__device__ __forceinline__ uint32_t render_pixel(const ResampleArgs& a, float p0, float p1, float p2, float pa,
                                                 uint32_t dst, const OutTables<LutF, LutB>& tb) {
    const LutF lut = tb.s2f;
    auto encode_channel = [&](const ResampleArgs& aa, float v) -> uint32_t { return ifhip::encode_channel<LIN>(aa, tb.l2s, v); };
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
// Cache policy of the canvas stores: non-temporal.  A canvas pixel is written once and never read again by the launch, and in
// a stream of reads mixed with 15 - 36 % of writes (the moderate ratios) the memory system moves 5.4 TB/s with `nt` stores
// against 4.9 with plain ones (tools/probes/stream_inflight_probe.hip).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_u32_untracked(uint32_t* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_f32x4_untracked(float4* p, float x, float y, float z, float w) {
    f32x4_t v = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <bool ALPHA, int LIN = -1, typename LutF, typename LutB>
__device__ __forceinline__ void store_pixel(const ResampleArgs& a, uint32_t img, uint32_t j, uint32_t u,
                                            float p0, float p1, float p2, float pa, const OutTables<LutF, LutB>& tb) {
    uint8_t* cp = a.canvas + static_cast<size_t>(img) * a.canvas_image_bytes
                  + static_cast<size_t>(a.y + j) * a.c_stride + static_cast<size_t>(a.x + u) * 4u;
    uint32_t* cw = reinterpret_cast<uint32_t*>(cp);           // canvas rows are 4-byte aligned (checked on host)
    uint32_t dst = 0;
    // BlendWithSelf reads the canvas pixel.  The load and its wait are one asm statement: a load the compiler tracks
    // would make it guard every later reuse of the destination register with s_waitcnt vmcnt(0) -- also on the paths
    // (other compositing modes) where no load was issued -- and each such wait drains the source rows in flight.
    if (ALPHA && a.mode == IFHIP_BLEND_WITH_SELF)
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(dst) : "v"(cw) : "memory");
    store_u32_untracked(cw, render_pixel<ALPHA, LIN>(a, p0, p1, p2, pa, dst, tb));
    if (a.f32_dump) {
        float4* d = reinterpret_cast<float4*>(a.f32_dump) + (static_cast<size_t>(img) * a.out_h + j) * a.out_w + u;
        store_f32x4_untracked(d, p0, p1, p2, ALPHA ? pa : 1.0f);
    }
}

}  // namespace ifhip
