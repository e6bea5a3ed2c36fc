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
    const float* h_wT;               // [h_max_taps][out_w], zero padded
    uint32_t h_max_taps;
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
    int linear;                      // working space
    int mode;                        // ifhip_compositing
    float m0, m1, m2, matte_a;       // matte colour in the working space, matte alpha / 255
    uint32_t n_images;
};

}  // namespace ifhip
