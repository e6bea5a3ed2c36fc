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
    // horizontal tables
    const uint32_t* h_left;
    const uint32_t* h_count;
    const float* h_wpad;             // [out_w][h_tpad], zero padded, rows 16-byte aligned
    uint32_t h_tpad;                 // taps per output rounded up to a multiple of 4
    uint32_t h_w_in_lds;             // 1: the strip's weight rows are staged in LDS
    // generic-kernel tables
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

// LDS carve of the fused kernel, shared by host (size) and device (offsets); all offsets in bytes, 16-aligned.
struct FusedLds {
    uint32_t lut, thr, hmeta, obuf, hw, inter, total;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline FusedLds fused_lds_layout(uint32_t n_u, uint32_t nquads, uint32_t tpad, int channels, bool w_in_lds) {
    FusedLds l;
    uint32_t off = 0;
    l.lut = off;   off += 256u * 32u * 4u;                 // sRGB->float table, one copy per LDS bank
    l.thr = off;   off += 256u * 2u;                       // linear->sRGB thresholds (binary search)
    l.hmeta = off; off += ((n_u * 8u) + 15u) & ~15u;
    l.obuf = off;  off += n_u * 16u;
    l.hw = off;    off += w_in_lds ? n_u * tpad * 4u : 0u;
    l.inter = off; off += nquads * 4u * static_cast<uint32_t>(channels) * 4u;
    l.total = off;
    return l;
}

}  // namespace ifhip
