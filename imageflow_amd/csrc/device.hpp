// device.hpp -- structures shared by the host launcher and the gfx950 kernels.
#pragma once
#include <cstddef>
#include <cstdint>

#include "common.hpp"

namespace ifhip {

struct Strip {          // one column strip of a fused launch
    uint32_t u0, u1;    // output columns [u0, u1)
    uint32_t cx0;       // first source column staged (multiple of 4)
    uint32_t nquads;    // number of 4-pixel groups staged: source columns [cx0, cx0 + 4*nquads)
};

// Everything a resample kernel needs; passed by value (lands in SGPRs / kernarg segment).
struct ResampleArgs {
    // source frames
    const uint8_t* in;
    size_t in_image_bytes;
    uint32_t in_stride, in_w, in_h;
    // planar YCbCr source (ycc != 0; the JPEG stage's component planes, decode fused into the row fetch): `in` is the Y
    // plane, in_stride the sample pitch and in_image_bytes the plane size shared by the three planes
    const uint8_t* in_cb;
    const uint8_t* in_cr;
    uint32_t ycc;
    // canvases
    uint8_t* canvas;
    size_t canvas_image_bytes;
    uint32_t c_stride, x, y, out_w, out_h;
    float* f32_dump;                 // nullable, [n][out_h][out_w][4]
    // vertical schedule (fused kernel)
    const VStep* steps;
    const uint32_t* band_begin;
    uint32_t n_bands;
    const Strip* strips;
    uint32_t n_strips;
    // horizontal tables (fused kernel): per output column {left, taps, first weight (float index into h_wu)}
    const uint32_t* h_meta2;         // fast horizontal pass (h_groups > 0): [out_w] first 4-column group | weight row id << 16
    uint32_t h_groups;               // G in 1..4: every output runs exactly G 4-tap groups (rows zero-padded to G groups,
                                     // h_wu holds them at a fixed pitch of G*4 floats); 0: per-output group counts (h_meta);
                                     // 16 + G2: G2 groups of TWO taps (h_meta2 counts 2-column groups, pitch G2*2 floats)
    const uint4* h_meta;             // [out_w] {first tap column rounded down to 4, 4-tap groups, weight row offset, taps valid in the last group}
    const float* h_wu;               // de-duplicated weight rows: (left & 3) leading zeros, taps, zero pad to 4; 16-B aligned
    uint32_t h_wu_floats;            // size of h_wu
    uint32_t h_w_in_lds;             // 1: h_wu is staged in LDS
    uint32_t l2s_in_lds;             // 1: the 16 KiB linear->sRGB table is staged in LDS (else threshold search)
    uint32_t lut_copies_log2;        // the sRGB->float table is replicated 2^n times in LDS (5: one copy per bank)
    uint32_t frames_per_wg;          // F: frames one workgroup works on side by side (narrow sources; tables shared)
    uint32_t lanes_per_frame;        // multiple of 64; the workgroup has F * lanes_per_frame lanes
    // generic-kernel tables
    const uint32_t* h_left;
    const uint32_t* h_count;
    const uint32_t* v_left;
    const uint32_t* v_count;
    const uint32_t* v_off;
    const float* v_w;
    const uint32_t* h_off;
    const float* h_w;
    // colour
    const float* lut_in;             // 256 floats in HBM: s2l (linear) or s2f (srgb)
    const uint8_t* l2s;              // 16384 bytes in HBM
    const uint16_t* l2s_thr;         // 256 u16 in HBM: thr[k] = first LUT index whose value is >= k+1 (65535 pad)
    int linear;                      // working space
    int mode;                        // ifhip_compositing
    float m0, m1, m2, matte_a;       // matte colour in the working space, matte alpha / 255
    uint32_t n_images;
};

// Launch parameters of the banded two-pass kernel (resample_kernels.hip; planned in api.cpp)
struct BandedArgs {
    uint32_t rows_per_band, n_bands, src_rows_cap;
    uint32_t frame_step;        // G: workgroup g of a band takes frames g, g + G, ...
    uint32_t h_w_floats;        // horizontal weights staged in LDS with flag 4: the whole table, or the widest strip's slice of it
    uint32_t flags;             // 1: short horizontal windows in registers; 2: window starts and ends ascend with the row (the band's
                                // source rows follow from its first and last row); 4: horizontal tables in LDS
    uint32_t n_strips, strip_w; // column strips of strip_w output columns each (wide frames: a band of whole rows would not fit
                                // the LDS); 1, out_w: whole rows
};

// Shape of the fused kernel for a ring of K rows and C channels per pixel.  The vertical accumulators alone take
// K*4*C registers per lane.  Three shapes, picked by a register estimate (checked against the compiler's report):
//   wide + pipelined : 1024 lanes (128 registers), D = 4 rows in flight, converted samples double buffered
//   wide + plain     : 1024 lanes, D = 4, no double buffering (register-heavier rings, e.g. K = 4 with alpha)
//   narrow           : big rings (K >= 6, or K >= 5 with alpha): 512 lanes (256 registers), pipelined; strips get narrower.
//                      (The same strips as 1024 lanes x 2 pixels with 8-byte loads -- twice the waves, half the accumulators
//                      per lane -- measured 1.3 % slower on cfg5: these shapes are instruction-bound; profiles/NOTEBOOK.md.)
// D is sized by bytes in flight: a CU needs ~46 KB outstanding to cover HBM latency at its share of the bandwidth
// (1024 lanes x 4 rows x 16 B = 64 KB; 512 lanes need 8 rows for the same).  Measured: cfg5 2.78 -> 2.46 ms with
// D = 8 on the narrow shape, cfg2 with alpha 2.05 -> 2.03 ms with D = 4 on the plain shape; round 3: 12 rows -1.3 %,
// 16 rows -2.0 % on cfg5 (216 / 232 registers; the step loop must then be unrolled past clang's pragma threshold,
// build.py passes -pragma-unroll-threshold) -- rings up to K = 6 take 16, the larger ones keep 8.
struct FusedShape { int threads, rows_in_flight, pipelined, px; };   // px: source pixels per lane (16- or 8-byte loads)
constexpr FusedShape fused_shape(int K, int channels) {
    // thresholds read off the compiler's register report (python -m imageflow_amd.kernel_report): no variant spills
    return (channels == 3 ? K <= 4 : K <= 2) ? FusedShape{1024, 4, 1, 4}
         : (channels == 3 ? K <= 5 : K <= 4) ? FusedShape{1024, 4, 0, 4}
                                             : FusedShape{512, K <= 6 ? 16 : 8, 1, 4};
}
constexpr int fused_max_threads(int K, int channels) { return fused_shape(K, channels).threads; }
constexpr int fused_max_quads(int K, int channels) { return fused_shape(K, channels).threads * fused_shape(K, channels).px / 4; }
// the step whose row the kernel requests while working on step i (see build_vschedule)
constexpr int fused_lookahead(int K, int channels) {
    return fused_shape(K, channels).pipelined ? fused_shape(K, channels).rows_in_flight + 1 : fused_shape(K, channels).rows_in_flight;
}

constexpr uint32_t fused_group_pitch(int channels) { return channels == 3 ? 48u : 80u; }   // bytes per 4-pixel group, fast pass

// LDS carve of the fused kernel, shared by host (size) and device (offsets); all offsets in bytes, 16-aligned.
struct FusedLds {
    uint32_t lut, thr, l2s, hmeta, obuf, hw, obuf_stride, inter, plane_pitch, inter_stride, total;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline FusedLds fused_lds_layout(uint32_t n_u, uint32_t nquads, uint32_t wu_floats, int channels, bool w_in_lds,
                                 bool l2s_in_lds, uint32_t lut_copies_log2, bool per_pixel, uint32_t frames,
                                 uint32_t fast_groups = 0) {
    FusedLds l;
    uint32_t off = 0;
    l.lut = off;   off += (256u << lut_copies_log2) * 4u;  // sRGB->float table, bank-interleaved copies
    l.thr = off;   off += 256u * 2u;                       // linear->sRGB thresholds (binary search fallback)
    l.l2s = off;   off += l2s_in_lds ? 16384u : 0u;        // linear->sRGB table
    l.hmeta = off; off += fast_groups ? ((n_u * 4u + 15u) & ~15u) : n_u * 16u;     // packed 4-byte records on the fast path
    l.obuf_stride = per_pixel ? 0u : 2u * n_u * 16u;       // horizontally filtered rows j-1 / j (per-chain mapping only)
    l.obuf = off;  off += frames * l.obuf_stride;          // one pair per frame slot
    l.hw = off;    off += w_in_lds ? ((wu_floats * 4u + 15u) & ~15u) : 0u;
    // vertically filtered row, C sub-planes of 16 B per 4-pixel group: sub-planes 0 / 1 hold the (c0, c1) pairs of
    // pixels 0,1 / 2,3 of the group (v_pk_fma_f32 operand order: one 16-byte read = two taps of two channels), the
    // rest hold c2 of the four pixels (no alpha) or the (c2, c3) pairs likewise.  One chunk per group and sub-plane
    // keeps the lane-to-lane stride of the horizontal gathers at (groups stepped) chunks, odd as often as even; the
    // pitch is == 4 (mod 64) dwords so that the sub-planes of one group sit 4 banks apart.
    l.plane_pitch = ((nquads * 4u + 63u) & ~63u) + 4u;                         // floats
    l.inter_stride = l.plane_pitch * static_cast<uint32_t>(channels) * 4u;    // bytes per buffered row
    if (fast_groups) {
        // Fast horizontal pass: the C 16-byte chunks of a 4-pixel group sit next to each other (group pitch 48 B, or
        // 80 B with alpha: 12 / 20 banks, so 8 consecutive groups cover all 32 banks), every read of an output is
        // base + immediate.  G - 1 groups past the staged columns are read with weight +0 and stay zero.
        l.plane_pitch = 0;
        l.inter_stride = ((nquads + fast_groups - 1u) * fused_group_pitch(channels) + 15u) & ~15u;
    }
    l.inter = off; off += frames * 2u * l.inter_stride;    // vertically filtered rows j / j+1, one pair per frame slot
    l.total = off;
    return l;
}

}  // namespace ifhip
