// jpeg_kernels.hip -- gfx950 kernels + C ABI for the JPEG pixel stage.
//
// Replaces the arithmetic libjpeg runs between entropy decoding and scanline output when imageflow decodes a JPEG
// (codecs/mozjpeg_decoder.rs:295-420 with dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE,
// out_color_space = JCS_EXT_BGRA): de-quantisation + 8x8 "islow" IDCT, triangle chroma up-sampling, fixed-point
// YCbCr -> BGRA.  All integer; bit-exact against oracle/jpeg_oracle.c (which is pinned to libjpeg-turbo's decode).
//
// Two kernels (bound: HBM; byte work, no MFMA):
//   jpeg_idct_kernel      8 lanes per 8x8 block, 32 blocks per workgroup.  A wave reads 8 consecutive blocks = 1 KiB of
//                         coefficients fully coalesced (16 B per lane), de-quantises into an LDS workspace (block
//                         pitch 72 dwords -> conflict-free column reads), runs the column pass (lane = column) and the
//                         row pass (lane = row) and stores 8 bytes per lane into the component plane.
//   jpeg_color_kernel     one lane = 4 adjacent output pixels x 16 rows; walking down, the lane keeps the last two chroma
//                         rows of the fancy up-sampler in registers (one new 4-byte chroma load per component and row
//                         pair), 16-byte BGRA stores.
// Algorithmic bytes per 4:2:0 pixel: 3 B coefficients + 4 B BGRA (+ 1.5 B written and re-read for the planes).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "common.hpp"

namespace ifhip {

typedef uint32_t uint4_nt __attribute__((ext_vector_type(4)));      // 16 bytes, usable with __builtin_nontemporal_load

struct JpegGeom {
    uint32_t width, height;
    int ncomp;
    uint32_t hs[3], vs[3];
    uint32_t hmax, vmax;
    uint32_t bw[3], bh[3];          // blocks per row / column (MCU padded)
    uint32_t pw[3], ph[3];          // plane width / height in samples (= 8 * blocks)
    uint32_t dw[3], dh[3];          // libjpeg downsampled_width / downsampled_height
    uint32_t blocks_before[4];      // prefix sums of bw*bh over components
    uint32_t out_w, out_h;          // ceil(width * scale_num / 8), ceil(height * scale_num / 8)
    uint32_t scale_num;             // 8 = full size; 1, 2, 4 = reduced-size decode
    uint32_t idct_n[3];             // samples per block edge each component's IDCT produces (1..6, 8, 10, 12)
    uint32_t luma_mode;             // 0: libjpeg's own (reduced) IDCT, 1: islow + flow_scale_spatial, 2: ... _srgb
    uint32_t upsample;              // 0 none (planes at output resolution), 1 h2v1 fancy, 2 h2v2 fancy, 3 h2v1 replicated,
                                    // 4 h2v2 replicated, 5 h1v2 fancy, 6 h1v2 replicated (jdsample.c jinit_upsampler)
};

struct JpegScalerTab {              // flow_scale_spatial tables for the luma size in use
    int32_t w[7][8];
    uint32_t log2_div[7];
    const uint16_t* s2l;
    const uint8_t* l2s;
};

struct JpegArgs {
    JpegGeom g;
    JpegScalerTab sc;
    const int16_t* coef[3];
    const uint16_t* qt;             // [n_images][ncomp][64]
    uint8_t* plane[3];              // [n_images][ph][pw]
    uint8_t* bgra;
    size_t image_bytes;
    uint32_t stride;
    uint32_t n_images;
    uint32_t comp;                  // component the IDCT launch works on (one launch per component)
};

// ---- IDCT ("islow": 13-bit fixed point, 12-multiply factorisation; ITU T.81 A.3.3 + the IJG constants) ----------
__device__ __forceinline__ int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ uint32_t range_limit(int32_t v) {   // libjpeg post-IDCT table, index (v & 1023)
    const uint32_t i = static_cast<uint32_t>(v) & 1023u;
    return i < 128u ? i + 128u : (i < 512u ? 255u : (i < 896u ? 0u : i - 896u));
}

// 32-bit integer multiplies run at a quarter of the rate of 24-bit ones on CDNA (v_mul_lo_u32 vs v_mul_i32_i24).  A 24-bit
// multiply returns the low 32 bits of the exact product, i.e. the same bits as the wrapping int32 multiply the oracle
// performs, whenever BOTH operands fit in 24 signed bits.  That always holds in the second (row) pass -- its inputs are
// first-pass results shifted right by 11, |w| <= 2^20, and the multiplicands are sums of at most four of them -- and in
// the first pass whenever every de-quantised coefficient of the block is below 2^21 (every 8-bit-precision file:
// |coefficient| <= 2^11 ... 2^15, quantiser <= 255); the kernels test that per wave and keep the 32-bit form for the rest.
template <bool M24>
__device__ __forceinline__ int32_t mulc(int32_t x, int32_t c) { return M24 ? __mul24(x, c) : x * c; }

// one 8-point pass; in[] are the 8 inputs, sh the descale amount; out via callback-free arrays
template <bool M24, bool RAW = false>
__device__ __forceinline__ void idct8(const int32_t (&in)[8], int32_t (&out)[8], int sh) {
    constexpr int32_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                      F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int32_t z2 = in[2], z3 = in[6];
    int32_t z1 = mulc<M24>(z2 + z3, F0_541);
    int32_t tmp2 = z1 + mulc<M24>(z3, -F1_847);
    int32_t tmp3 = z1 + mulc<M24>(z2, F0_765);
    // descale()'s rounding constant goes into the even part once (one shift-add each) instead of into the eight outputs:
    // int32 addition wraps the same way in any order, so every sum below is the oracle's plus 2^(sh-1), bit for bit
    const uint32_t half = 1u << (sh - 1);
    int32_t tmp0 = static_cast<int32_t>((static_cast<uint32_t>(in[0] + in[4]) << 13) + half);
    int32_t tmp1 = static_cast<int32_t>((static_cast<uint32_t>(in[0] - in[4]) << 13) + half);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = mulc<M24>(z3 + z4, F1_175);
    tmp0 = mulc<M24>(tmp0, F0_298); tmp1 = mulc<M24>(tmp1, F2_053); tmp2 = mulc<M24>(tmp2, F3_072); tmp3 = mulc<M24>(tmp3, F1_501);
    z1 = mulc<M24>(z1, -F0_899); z2 = mulc<M24>(z2, -F2_562); z3 = mulc<M24>(z3, -F1_961); z4 = mulc<M24>(z4, -F0_390);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    if (RAW) sh = 0;                                      // the caller takes its bits out of the sums itself
    out[0] = (tmp10 + tmp3) >> sh; out[7] = (tmp10 - tmp3) >> sh;
    out[1] = (tmp11 + tmp2) >> sh; out[6] = (tmp11 - tmp2) >> sh;
    out[2] = (tmp12 + tmp1) >> sh; out[5] = (tmp12 - tmp1) >> sh;
    out[3] = (tmp13 + tmp0) >> sh; out[4] = (tmp13 - tmp0) >> sh;
}
// first (column) pass: 24-bit multiplies when the whole wave's blocks are in range (wave-uniform choice)
__device__ __forceinline__ void idct8_pass1(const int32_t (&in)[8], int32_t (&out)[8], bool small) {
    if (small) idct8<true>(in, out, 11);
    else idct8<false>(in, out, 11);
}
__device__ __forceinline__ bool wave_all_small(const int32_t (&d)[8], bool lane_on) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) m |= static_cast<uint32_t>(d[k] < 0 ? -d[k] : d[k]);
    return __all(!lane_on || m < (1u << 21)) != 0;
}

// reduced-size passes of jidctred.c (jpeg_idct_4x4 / jpeg_idct_2x2): same 13-bit constants family
template <bool M24>
__device__ __forceinline__ void idct4_pass(int32_t d0, int32_t d1, int32_t d2, int32_t d3, int32_t d5, int32_t d6, int32_t d7,
                                           int32_t (&out)[4], int sh) {
    const int32_t t0 = static_cast<int32_t>((static_cast<uint32_t>(d0) << 14) + (1u << (sh - 1)));     // rounding folded as in idct8
    const int32_t t2 = mulc<M24>(d2, 15137) + mulc<M24>(d6, -6270);
    const int32_t t10 = t0 + t2, t12 = t0 - t2;
    const int32_t o0 = mulc<M24>(d7, -1730) + mulc<M24>(d5, 11893) + mulc<M24>(d3, -17799) + mulc<M24>(d1, 8697);
    const int32_t o2 = mulc<M24>(d7, -4176) + mulc<M24>(d5, -4926) + mulc<M24>(d3, 7373) + mulc<M24>(d1, 20995);
    out[0] = (t10 + o2) >> sh; out[3] = (t10 - o2) >> sh;
    out[1] = (t12 + o0) >> sh; out[2] = (t12 - o0) >> sh;
}
template <bool M24>
__device__ __forceinline__ void idct2_pass(int32_t d0, int32_t d1, int32_t d3, int32_t d5, int32_t d7, int32_t (&out)[2], int sh) {
    const int32_t t10 = static_cast<int32_t>((static_cast<uint32_t>(d0) << 15) + (1u << (sh - 1)));
    const int32_t t0 = mulc<M24>(d7, -5906) + mulc<M24>(d5, 6967) + mulc<M24>(d3, -10426) + mulc<M24>(d1, 29692);
    out[0] = (t10 + t0) >> sh; out[1] = (t10 - t0) >> sh;
}

// libjpeg's other scaled IDCTs (jidctint.c jpeg_idct_3x3 / 5x5 / 6x6 / 10x10 / 12x12): the block routines behind
// scale_num 3, 5, 6 (luma NxN; 2x2 sub-sampled chroma 2N x 2N).  One N-point pass each; `dc` arrives shifted left by
// CONST_BITS with the pass's rounding constant added, PASS1 selects the first-pass forms (a few outputs combine
// separately shifted halves there).  Plain arithmetic shifts, as the library.
#define IFHIP_FIXC(x) static_cast<int32_t>((x) * 8192.0 + 0.5)
template <bool PASS1>
__device__ __forceinline__ void idct3_pass(int32_t dc, int32_t d1, int32_t d2, int32_t (&o)[3]) {
    constexpr int sh = PASS1 ? 11 : 18;
    const int32_t t12 = d2 * IFHIP_FIXC(0.707106781);
    const int32_t t10 = dc + t12, t2 = dc - t12 - t12;
    const int32_t t0 = d1 * IFHIP_FIXC(1.224744871);
    o[0] = (t10 + t0) >> sh; o[2] = (t10 - t0) >> sh; o[1] = t2 >> sh;
}
template <bool PASS1>
__device__ __forceinline__ void idct5_pass(int32_t dc, int32_t d1, int32_t d2, int32_t d3, int32_t d4, int32_t (&o)[5]) {
    constexpr int sh = PASS1 ? 11 : 18;
    int32_t z1 = (d2 + d4) * IFHIP_FIXC(0.790569415);
    const int32_t z2 = (d2 - d4) * IFHIP_FIXC(0.353553391);
    const int32_t z3 = dc + z2;
    const int32_t t10 = z3 + z1, t11 = z3 - z1;
    const int32_t t12 = dc - static_cast<int32_t>(static_cast<uint32_t>(z2) << 2);
    z1 = (d1 + d3) * IFHIP_FIXC(0.831253876);
    const int32_t t0 = z1 + d1 * IFHIP_FIXC(0.513743148);
    const int32_t t1 = z1 - d3 * IFHIP_FIXC(2.176250899);
    o[0] = (t10 + t0) >> sh; o[4] = (t10 - t0) >> sh; o[1] = (t11 + t1) >> sh; o[3] = (t11 - t1) >> sh; o[2] = t12 >> sh;
}
template <bool PASS1>
__device__ __forceinline__ void idct6_pass(int32_t dc, int32_t d1, int32_t d2, int32_t d3, int32_t d4, int32_t d5, int32_t (&o)[6]) {
    constexpr int sh = PASS1 ? 11 : 18;
    int32_t t10 = d4 * IFHIP_FIXC(0.707106781);
    const int32_t t1a = dc + t10;
    const int32_t t11 = dc - t10 - t10;
    const int32_t t0a = d2 * IFHIP_FIXC(1.224744871);
    t10 = t1a + t0a;
    const int32_t t12 = t1a - t0a;
    const int32_t t1 = (d1 + d5) * IFHIP_FIXC(0.366025404);
    const int32_t t0 = t1 + static_cast<int32_t>(static_cast<uint32_t>(d1 + d3) << 13);
    const int32_t t2 = t1 + static_cast<int32_t>(static_cast<uint32_t>(d5 - d3) << 13);
    o[0] = (t10 + t0) >> sh; o[5] = (t10 - t0) >> sh;
    o[2] = (t12 + t2) >> sh; o[3] = (t12 - t2) >> sh;
    if (PASS1) {
        const int32_t m = static_cast<int32_t>(static_cast<uint32_t>(d1 - d3 - d5) << 2), h = t11 >> sh;
        o[1] = h + m; o[4] = h - m;
    } else {
        const int32_t m = static_cast<int32_t>(static_cast<uint32_t>(d1 - d3 - d5) << 13);
        o[1] = (t11 + m) >> sh; o[4] = (t11 - m) >> sh;
    }
}
template <bool PASS1>
__device__ __forceinline__ void idct10_pass(int32_t dc, const int32_t (&d)[8], int32_t (&o)[10]) {
    constexpr int sh = PASS1 ? 11 : 18;
    int32_t z1 = d[4] * IFHIP_FIXC(1.144122806), z2 = d[4] * IFHIP_FIXC(0.437016024);
    int32_t t10 = dc + z1, t11 = dc - z2;
    const int32_t t22 = dc - static_cast<int32_t>(static_cast<uint32_t>(z1 - z2) << 1);
    z1 = (d[2] + d[6]) * IFHIP_FIXC(0.831253876);
    int32_t t12 = z1 + d[2] * IFHIP_FIXC(0.513743148);
    int32_t t13 = z1 - d[6] * IFHIP_FIXC(2.176250899);
    const int32_t t20 = t10 + t12, t24 = t10 - t12, t21 = t11 + t13, t23 = t11 - t13;
    const int32_t o5s = static_cast<int32_t>(static_cast<uint32_t>(d[5]) << 13);
    t11 = d[3] + d[7];
    t13 = d[3] - d[7];
    t12 = t13 * IFHIP_FIXC(0.309016994);
    z2 = t11 * IFHIP_FIXC(0.951056516);
    int32_t z4 = o5s + t12;
    t10 = d[1] * IFHIP_FIXC(1.396802247) + z2 + z4;
    const int32_t t14 = d[1] * IFHIP_FIXC(0.221231742) - z2 + z4;
    z2 = t11 * IFHIP_FIXC(0.587785252);
    z4 = o5s - t12 - static_cast<int32_t>(static_cast<uint32_t>(t13) << 12);
    const int32_t mid = d[1] - t13;
    t11 = d[1] * IFHIP_FIXC(1.260073511) - z2 - z4;
    t13 = d[1] * IFHIP_FIXC(0.642039522) - z2 + z4;
    o[0] = (t20 + t10) >> sh; o[9] = (t20 - t10) >> sh;
    o[1] = (t21 + t11) >> sh; o[8] = (t21 - t11) >> sh;
    o[3] = (t23 + t13) >> sh; o[6] = (t23 - t13) >> sh;
    o[4] = (t24 + t14) >> sh; o[5] = (t24 - t14) >> sh;
    if (PASS1) {
        const int32_t a = t22 >> sh, b = static_cast<int32_t>(static_cast<uint32_t>(mid - d[5]) << 2);
        o[2] = a + b; o[7] = a - b;
    } else {
        const int32_t b = static_cast<int32_t>(static_cast<uint32_t>(mid) << 13) - o5s;
        o[2] = (t22 + b) >> sh; o[7] = (t22 - b) >> sh;
    }
}
template <bool PASS1>
__device__ __forceinline__ void idct12_pass(int32_t dc, const int32_t (&d)[8], int32_t (&o)[12]) {
    constexpr int sh = PASS1 ? 11 : 18;
    int32_t z4 = d[4] * IFHIP_FIXC(1.224744871);
    int32_t t10 = dc + z4, t11 = dc - z4;
    z4 = d[2] * IFHIP_FIXC(1.366025404);
    int32_t z1 = static_cast<int32_t>(static_cast<uint32_t>(d[2]) << 13);
    int32_t z2 = static_cast<int32_t>(static_cast<uint32_t>(d[6]) << 13);
    int32_t t12 = z1 - z2;
    const int32_t t21 = dc + t12, t24 = dc - t12;
    t12 = z4 + z2;
    const int32_t t20 = t10 + t12, t25 = t10 - t12;
    t12 = z4 - z1 - z2;
    const int32_t t22 = t11 + t12, t23 = t11 - t12;
    z1 = d[1]; z2 = d[3];
    int32_t z3 = d[5];
    z4 = d[7];
    t11 = z2 * IFHIP_FIXC(1.306562965);
    int32_t t14 = z2 * (-4433);
    t10 = z1 + z3;
    int32_t t15 = (t10 + z4) * IFHIP_FIXC(0.860918669);
    t12 = t15 + t10 * IFHIP_FIXC(0.261052384);
    t10 = t12 + t11 + z1 * IFHIP_FIXC(0.280143716);
    int32_t t13 = (z3 + z4) * (-IFHIP_FIXC(1.045510580));
    t12 += t13 + t14 - z3 * IFHIP_FIXC(1.478575242);
    t13 += t15 - t11 + z4 * IFHIP_FIXC(1.586706681);
    t15 += t14 - z1 * IFHIP_FIXC(0.676326758) - z4 * IFHIP_FIXC(1.982889723);
    z1 -= z4;
    z2 -= z3;
    z3 = (z1 + z2) * 4433;
    t11 = z3 + z1 * 6270;
    t14 = z3 - z2 * 15137;
    o[0] = (t20 + t10) >> sh; o[11] = (t20 - t10) >> sh;
    o[1] = (t21 + t11) >> sh; o[10] = (t21 - t11) >> sh;
    o[2] = (t22 + t12) >> sh; o[9] = (t22 - t12) >> sh;
    o[3] = (t23 + t13) >> sh; o[8] = (t23 - t13) >> sh;
    o[4] = (t24 + t14) >> sh; o[7] = (t24 - t14) >> sh;
    o[5] = (t25 + t15) >> sh; o[6] = (t25 - t15) >> sh;
}
__device__ __forceinline__ int32_t dc_pass1(int32_t d0) { return static_cast<int32_t>(static_cast<uint32_t>(d0) << 13) + (1 << 10); }
__device__ __forceinline__ int32_t dc_pass2(int32_t w0) { return static_cast<int32_t>(static_cast<uint32_t>(w0 + 16) << 13); }

constexpr int kBlocksPerWg = 32;
constexpr int kBlockPitch = 72;     // dwords per 8x8 workspace in LDS (64 + 8: spreads 4 blocks over the 32 banks)
constexpr int kBigPitch = 104;      // jpeg_idct_kernel: up to 12 workspace rows of 8 (the 10x10 / 12x12 IDCTs) + 8

__global__ void __launch_bounds__(256) jpeg_idct_kernel(const JpegArgs a) {
    __shared__ __attribute__((aligned(16))) int32_t ws[kBlocksPerWg * kBigPitch];
    // grid: x = groups of 32 blocks of component a.comp, y = image
    const uint32_t t = threadIdx.x, lane8 = t & 7u, lb = t >> 3;
    const uint32_t c = a.comp + blockIdx.z, img = blockIdx.y;          // blockIdx.z: the two chroma components in one launch
    const uint32_t bidx = blockIdx.x * kBlocksPerWg + lb;
    const bool on = bidx < a.g.bw[c] * a.g.bh[c];
    const uint32_t n = a.g.idct_n[c];                                   // output samples per block edge
    const bool spatial = (c == 0u) && a.g.luma_mode != 0u && n < 8u;    // islow, then imageflow's block scaler
    const uint32_t m = spatial ? 8u : n;                                // size of the IDCT actually run
    int32_t* w = ws + lb * kBigPitch;
    if (on) {
        const uint32_t nblk = a.g.bw[c] * a.g.bh[c];
        const int16_t* src = a.coef[c] + (static_cast<size_t>(img) * nblk + bidx) * 64u + lane8 * 8u;
        // the luma scalers' islow runs with the FIRST component's table, as jpeg_idct_islow(cinfo, compptr, ...) does
        const uint16_t* q = a.qt + (static_cast<size_t>(img) * a.g.ncomp + c) * 64u + lane8 * 8u;
        const uint4 cv = *reinterpret_cast<const uint4*>(src);          // row lane8 of the block: 8 x int16
        const uint4 qv = *reinterpret_cast<const uint4*>(q);
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int32_t c0 = static_cast<int16_t>(cw[k] & 0xffffu), c1 = static_cast<int16_t>(cw[k] >> 16);
            const int32_t q0 = static_cast<int32_t>(qw[k] & 0xffffu), q1 = static_cast<int32_t>(qw[k] >> 16);
            w[lane8 * 8u + 2 * k] = __mul24(c0, q0);            // 16-bit operands: exact
            w[lane8 * 8u + 2 * k + 1] = __mul24(c1, q1);
        }
    }
    __syncthreads();
    int32_t col_in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on) {
#pragma unroll
        for (int r = 0; r < 8; ++r) col_in[r] = w[r * 8 + lane8];
    }
    const bool small = wave_all_small(col_in, on);
    if (on) {                                   // column pass: lane8 = column
        int32_t (&in)[8] = col_in;
        if (m == 8u) {
            int32_t out[8];
            idct8_pass1(in, out, small);        // CONST_BITS - PASS1_BITS
#pragma unroll
            for (int r = 0; r < 8; ++r) w[r * 8 + lane8] = out[r];
        } else if (m == 4u) {
            int32_t out[4];
            if (small) idct4_pass<true>(in[0], in[1], in[2], in[3], in[5], in[6], in[7], out, 12);     // CONST_BITS - PASS1_BITS + 1
            else idct4_pass<false>(in[0], in[1], in[2], in[3], in[5], in[6], in[7], out, 12);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r * 8 + lane8] = out[r];
        } else if (m == 2u) {
            int32_t out[2];
            if (small) idct2_pass<true>(in[0], in[1], in[3], in[5], in[7], out, 13);                   // CONST_BITS - PASS1_BITS + 2
            else idct2_pass<false>(in[0], in[1], in[3], in[5], in[7], out, 13);
            w[lane8] = out[0]; w[8 + lane8] = out[1];
        } else if (m == 3u) {
            if (lane8 < 3u) { int32_t out[3]; idct3_pass<true>(dc_pass1(in[0]), in[1], in[2], out); for (int r = 0; r < 3; ++r) w[r * 8 + lane8] = out[r]; }
        } else if (m == 5u) {
            if (lane8 < 5u) { int32_t out[5]; idct5_pass<true>(dc_pass1(in[0]), in[1], in[2], in[3], in[4], out); for (int r = 0; r < 5; ++r) w[r * 8 + lane8] = out[r]; }
        } else if (m == 6u) {
            if (lane8 < 6u) { int32_t out[6]; idct6_pass<true>(dc_pass1(in[0]), in[1], in[2], in[3], in[4], in[5], out); for (int r = 0; r < 6; ++r) w[r * 8 + lane8] = out[r]; }
        } else if (m == 10u) {
            int32_t out[10]; idct10_pass<true>(dc_pass1(in[0]), in, out);
#pragma unroll
            for (int r = 0; r < 10; ++r) w[r * 8 + lane8] = out[r];
        } else if (m == 12u) {
            int32_t out[12]; idct12_pass<true>(dc_pass1(in[0]), in, out);
#pragma unroll
            for (int r = 0; r < 12; ++r) w[r * 8 + lane8] = out[r];
        }
    }
    __syncthreads();
    uint8_t* plane = a.plane[c] + static_cast<size_t>(img) * a.g.pw[c] * a.g.ph[c];
    const uint32_t by = on ? bidx / a.g.bw[c] : 0u, bx = on ? bidx - by * a.g.bw[c] : 0u;
    uint8_t* bytes = reinterpret_cast<uint8_t*>(w);          // the 8x8 bytes of a full IDCT, for the spatial scaler
    if (on) {                                   // row pass: lane8 = row
        if (m == 8u) {
            int32_t in[8], out[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) in[k] = w[lane8 * 8u + k];
            idct8<true>(in, out, 18);           // CONST_BITS + PASS1_BITS + 3 (second pass: always in 24-bit range)
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo |= range_limit(out[k]) << (8 * k);
                hi |= range_limit(out[4 + k]) << (8 * k);
            }
            if (!spatial) {
                uint8_t* dst = plane + static_cast<size_t>(by * 8u + lane8) * a.g.pw[c] + bx * 8u;
                *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
            } else {
                // keep the row's 8 bytes in the first two dwords of the lane's own (already consumed) workspace row
                *reinterpret_cast<uint2*>(w + lane8 * 8u) = make_uint2(lo, hi);
            }
        } else if (m == 4u) {
            if (lane8 < 4u) {
                const int32_t* r = w + lane8 * 8u;
                int32_t out[4];
                idct4_pass<true>(r[0], r[1], r[2], r[3], r[5], r[6], r[7], out, 19);       // CONST_BITS + PASS1_BITS + 3 + 1 (|r| <= 2^20: 24-bit exact)
                const uint32_t v = range_limit(out[0]) | (range_limit(out[1]) << 8) | (range_limit(out[2]) << 16) | (range_limit(out[3]) << 24);
                *reinterpret_cast<uint32_t*>(plane + static_cast<size_t>(by * 4u + lane8) * a.g.pw[c] + bx * 4u) = v;
            }
        } else if (m == 2u) {
            if (lane8 < 2u) {
                const int32_t* r = w + lane8 * 8u;
                int32_t out[2];
                idct2_pass<true>(r[0], r[1], r[3], r[5], r[7], out, 20);                    // CONST_BITS + PASS1_BITS + 3 + 2
                const uint32_t v = range_limit(out[0]) | (range_limit(out[1]) << 8);
                *reinterpret_cast<uint16_t*>(plane + static_cast<size_t>(by * 2u + lane8) * a.g.pw[c] + bx * 2u) = static_cast<uint16_t>(v);
            }
        } else if (m == 1u) {                                                        // 1x1: DC / 8
            if (lane8 == 0u) plane[static_cast<size_t>(by) * a.g.pw[c] + bx] = static_cast<uint8_t>(range_limit(descale(w[0], 3)));
        } else {                                                                     // 3, 5, 6, 10, 12: byte stores (parity path)
            for (uint32_t row = lane8; row < m; row += 8u) {
                const int32_t* r = w + row * 8u;
                int32_t out[12];
                if (m == 3u) { int32_t o[3]; idct3_pass<false>(dc_pass2(r[0]), r[1], r[2], o); for (int k = 0; k < 3; ++k) out[k] = o[k]; }
                else if (m == 5u) { int32_t o[5]; idct5_pass<false>(dc_pass2(r[0]), r[1], r[2], r[3], r[4], o); for (int k = 0; k < 5; ++k) out[k] = o[k]; }
                else if (m == 6u) { int32_t o[6]; idct6_pass<false>(dc_pass2(r[0]), r[1], r[2], r[3], r[4], r[5], o); for (int k = 0; k < 6; ++k) out[k] = o[k]; }
                else {
                    int32_t d[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = r[k];
                    if (m == 10u) { int32_t o[10]; idct10_pass<false>(dc_pass2(r[0]), d, o); for (int k = 0; k < 10; ++k) out[k] = o[k]; }
                    else { idct12_pass<false>(dc_pass2(r[0]), d, out); }
                }
                uint8_t* dst = plane + static_cast<size_t>(by * m + row) * a.g.pw[c] + bx * m;
                for (uint32_t k = 0; k < m; ++k) dst[k] = static_cast<uint8_t>(range_limit(out[k]));
            }
        }
    }
    if (spatial) {
        // flow_scale_spatial[_srgb]_NxN on the block's 8x8 bytes (codecs_jpeg_idct_fast.c): rows are combined with the
        // integer weights of output row r, then columns with those of output column cc, rounded by the two divisors'
        // shift; the _srgb forms do it in 12-bit linear light (two lookup tables).  All eight lanes of the block work:
        //   lane = source COLUMN j: its 8 bytes -> linear (8 lookups, not 8 per output row), V[r][j] for r < n with the
        //   weights on the scalar path;  then lane = output (r, cc), ceil(n*n / 8) each: 8 multiply-adds over V[r][.];
        //   then lane = output row r: one store of n bytes.
        // |weight| <= 117, linear <= 4095, sums of weights <= 512: every product fits 24 x 24 -> 32 bits.
        // The 8 lanes of a block sit in one wave, whose LDS operations execute in program order: no barrier between
        // the phases, only the compiler has to keep the order (it cannot prove the accesses distinct).
        __shared__ uint16_t s2l_lds[256];
        __shared__ uint8_t l2s_lds[4096];
        __shared__ __attribute__((aligned(16))) int32_t wts_lds[64];     // [cc][j] weights, [56 + r] log2 divisors
        const bool srgb = a.g.luma_mode == 2u;
        if (srgb) {
            s2l_lds[t] = a.sc.s2l[t];
            reinterpret_cast<uint4*>(l2s_lds)[t] = reinterpret_cast<const uint4*>(a.sc.l2s)[t];
        }
        if (t < 56u) wts_lds[t] = a.sc.w[t >> 3][t & 7u];
        else if (t < 63u) wts_lds[t] = static_cast<int32_t>(a.sc.log2_div[t - 56u]);
        __syncthreads();
        if (on) {
            int32_t lin[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t byte = bytes[i * 32 + lane8];
                lin[i] = srgb ? static_cast<int32_t>(s2l_lds[byte]) : static_cast<int32_t>(byte);
            }
            asm volatile("" ::: "memory");                   // the byte reads stay in front of the stores below
#pragma unroll
            for (uint32_t r = 0; r < 7u; ++r)
                if (r < n) {
                    int32_t acc = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc += __mul24(a.sc.w[r][i], lin[i]);
                    w[r * 8u + lane8] = acc;                 // V[r][j]
                }
            asm volatile("" ::: "memory");
            uint8_t* otile = reinterpret_cast<uint8_t*>(w + 64);          // n x n output bytes (the workspace has 104 dwords)
            const uint32_t inv_n = (65536u + n - 1u) / n;                 // o / n for o < 49: (o * ceil(2^16 / n)) >> 16
            for (uint32_t o = lane8; o < n * n; o += 8u) {
                const uint32_t r = (o * inv_n) >> 16, cc = o - r * n;
                const int4 v0 = *reinterpret_cast<const int4*>(w + r * 8u), v1 = *reinterpret_cast<const int4*>(w + r * 8u + 4u);
                const int4 g0 = *reinterpret_cast<const int4*>(wts_lds + cc * 8u), g1 = *reinterpret_cast<const int4*>(wts_lds + cc * 8u + 4u);
                const uint32_t sh = static_cast<uint32_t>(wts_lds[56u + r] + wts_lds[56u + cc]);
                int32_t sum = static_cast<int32_t>(1u << (sh - 1u));
                sum = (__mul24(v0.x, g0.x) + sum); sum = (__mul24(v0.y, g0.y) + sum); sum = (__mul24(v0.z, g0.z) + sum); sum = (__mul24(v0.w, g0.w) + sum);
                sum = (__mul24(v1.x, g1.x) + sum); sum = (__mul24(v1.y, g1.y) + sum); sum = (__mul24(v1.z, g1.z) + sum); sum = (__mul24(v1.w, g1.w) + sum);
                uint32_t ob;
                if (sum < 0) ob = 0;
                else if (static_cast<uint32_t>(sum) >= (4096u << sh)) ob = 255;
                else ob = srgb ? l2s_lds[sum >> sh] : static_cast<uint32_t>(sum >> sh);
                otile[o] = static_cast<uint8_t>(ob);
            }
            asm volatile("" ::: "memory");
            if (lane8 < n) {
                uint8_t* orow = plane + static_cast<size_t>(by * n + lane8) * a.g.pw[c] + bx * n;
                const uint8_t* src = otile + lane8 * n;
                if (n == 4u) *reinterpret_cast<uint32_t*>(orow) = *reinterpret_cast<const uint32_t*>(src);        // (plane pitch and bx * 4: 4-byte aligned)
                else if (n == 2u) *reinterpret_cast<uint16_t*>(orow) = *reinterpret_cast<const uint16_t*>(src);
                else for (uint32_t cc = 0; cc < n; ++cc) orow[cc] = src[cc];
            }
        }
    }
}

__device__ __forceinline__ void wave_sync_lds() {       // lanes of ONE wave hand data over through LDS: order its LDS traffic, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- one lane = one 8x8 block --------------------------------------------------------------------------------------------
// The common block routines -- islow 8x8 (MODE 0), jidctred's 4x4 (MODE 1), islow followed by imageflow's
// flow_scale_spatial[_srgb]_NxN (MODE 2) -- with the whole block in the registers of ONE lane: both passes run on
// compile-time register indices, nothing goes through LDS, no wave-scope ordering, and the per-lane overhead of the
// eight-lanes-per-block form (de-quantising 8 values, 24 LDS accesses, index arithmetic, packing -- about 190 of its 250
// instructions per lane, i.e. 1 500 of 2 000 per block) is paid once per block instead of eight times.  A wave reads 64
// consecutive blocks = 8 KiB (every line is consumed by the lane's eight 16-byte loads) and stores 512 contiguous bytes
// per output row.  Same integer arithmetic as jpeg_idct_kernel (which keeps the rarer sizes 1, 2, 3, 5, 6, 10, 12).
// flow_scale_spatial weights as compile-time data (the same rows block_scalers.cpp serves to the host and, through
// JpegScalerTab, to the eight-lanes kernel): with the block size N a template parameter every multiply by a zero weight --
// 18 of the 32 weights at N = 4 -- disappears, the others become literals.
struct ScalerRowC { int n, r, log2_div; int w[8]; };
__device__ constexpr ScalerRowC kScalerRowsC[] = {
#include "block_scaler_weights.inc"
};
constexpr int scaler_row_index(int n, int r) {
    for (int i = 0; i < static_cast<int>(sizeof(kScalerRowsC) / sizeof(kScalerRowsC[0])); ++i)
        if (kScalerRowsC[i].n == n && kScalerRowsC[i].r == r) return i;
    return 0;
}

__device__ __forceinline__ uint32_t range_limit_fast(int32_t v) {    // == range_limit(v) for v in [-384, 383]
    const int32_t x = v + 128;
    return static_cast<uint32_t>(x < 0 ? 0 : (x > 255 ? 255 : x));        // (v_med3_i32)
}

template <bool M24>
__device__ __forceinline__ void bpl_column_pass(int32_t (&ws)[8][8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int32_t in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            in[r] = ws[r][k];
            // (the 32-bit form is the rare one: hidden from value numbering, or the sums it has in common with the 24-bit
            // form are hoisted above the wave's choice for all eight columns at once -- 64 live values, spilled)
            if (!M24) asm volatile("" : "+v"(in[r]));
        }
        idct8<M24>(in, out, 11);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][k] = out[r];
        __builtin_amdgcn_sched_barrier(0);                      // one column at a time: interleaved, the eight passes' temporaries spill
    }
}
// flow_scale_spatial[_srgb]_NxN on one block of (linear) samples.  SRGB: the result goes back through the 4 096-entry
// linear -> sRGB table, whose ends are 0 and 255 (block_scalers.cpp checks), so the reference's two range tests are a clamp
// of the index.
// The scaler's multiply-adds, spelled as the instruction.  Left to the optimiser a 24-bit multiply by a weight gets
// rewritten -- factored into weight * (a + b), or, once an operand's range is known, into a plain 32-bit multiply the
// instruction selector cannot prove narrow again -- and comes out as v_mul_lo_u32 / v_mad_u64_u32, which issue at a
// quarter of the rate (25 of them per block in the 4x4 scaler).  w is a compile-time weight (scalar operand).
__device__ __forceinline__ int32_t mad24(int32_t x, int32_t w, int32_t acc) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ int32_t mul24c(int32_t x, int32_t w) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, 0" : "=v"(d) : "v"(x), "s"(w));
    return d;
}

template <int N, bool SRGB>
__device__ __forceinline__ void bpl_scale_block(const int32_t (&lin)[8][8], uint8_t* plane, uint32_t by, uint32_t bx, uint32_t pitch,
                                                const uint8_t* l2s_lds) {
    constexpr uint32_t n = N;
    constexpr int base = scaler_row_index(N, 0);                // the N rows of size N are consecutive in the table
#pragma unroll
    for (uint32_t r = 0; r < n; ++r) {
        int32_t V[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int32_t acc = 0;
            bool first = true;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (kScalerRowsC[base + r].w[i] != 0) {
                    acc = first ? mul24c(lin[i][j], kScalerRowsC[base + r].w[i]) : mad24(lin[i][j], kScalerRowsC[base + r].w[i], acc);
                    first = false;
                }
            V[j] = acc;
        }
        uint32_t packed = 0;
        uint8_t* orow = plane + static_cast<size_t>(by * n + r) * pitch + bx * n;
#pragma unroll
        for (uint32_t cc = 0; cc < n; ++cc) {
            const uint32_t sh = static_cast<uint32_t>(kScalerRowsC[base + r].log2_div + kScalerRowsC[base + cc].log2_div);
            int32_t sum = static_cast<int32_t>(1u << (sh - 1u));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kScalerRowsC[base + cc].w[j] != 0) sum = mad24(V[j], kScalerRowsC[base + cc].w[j], sum);
            const int32_t q = sum >> sh;                        // (sum < 0 <=> q < 0; sum >= 4096 << sh <=> q >= 4096)
            uint32_t ob;
            if (SRGB) ob = l2s_lds[q < 0 ? 0 : (q > 4095 ? 4095 : q)];
            else ob = q < 0 ? 0u : (q > 4095 ? 255u : static_cast<uint32_t>(q) & 255u);
            if (n == 4u || n == 2u) packed |= ob << (8u * cc);
            else orow[cc] = static_cast<uint8_t>(ob);
        }
        if (n == 4u) *reinterpret_cast<uint32_t*>(orow) = packed;               // (plane pitch and bx * 4: 4-byte aligned)
        else if (n == 2u) *reinterpret_cast<uint16_t*>(orow) = static_cast<uint16_t>(packed);
    }
}

// lanes per workgroup: with the scaler's tables 8 waves (2 workgroups x 78 KiB of LDS = 4 waves per SIMD; 4-wave workgroups of
// 41.5 KiB fit only three times), the plain routines 4 waves (4 x 38 KiB)
constexpr uint32_t bpl_threads(int mode) { return mode == 2 ? 512u : 256u; }
typedef uint4_nt (*BplStage)[64 * 9];
template <int MODE, int N>
__device__ __forceinline__ void idct_block_per_lane(const JpegArgs& a, uint32_t c, uint32_t img, uint32_t wg, BplStage stage,
                                                    const uint32_t* lim_lds, const uint8_t* l2s_lds, const uint8_t* lim8_lds) {
    const uint32_t t = threadIdx.x;
    const bool srgb = MODE == 2 && a.g.luma_mode == 2u;
    const uint32_t nblk = a.g.bw[c] * a.g.bh[c];
    const uint32_t bidx = wg * bpl_threads(MODE) + t;
    // A wave's 64 blocks are 8 KiB of consecutive coefficients: read them with 8 fully coalesced 16-byte loads per lane and
    // hand each lane its own block through LDS (block pitch 9 x 16 B: the 16-byte row reads of neighbouring lanes fall 4
    // banks apart).  Lanes reading their blocks straight from global memory -- 16 bytes of 64 different cache lines per
    // instruction -- made this kernel 1.6x slower than the eight-lanes form it replaces (profiles/r3_jpeg_kernels_*.txt).
    const uint32_t wv = t >> 6, ln = t & 63u;
    const uint32_t wave_block0 = wg * bpl_threads(MODE) + wv * 64u;
    if (wave_block0 >= nblk) return;                           // (whole wave: nothing below is a workgroup barrier)
    const uint32_t wave_vecs = min(64u, nblk - wave_block0) * 8u;    // 16-byte rows this wave owns
    const uint4_nt* wsrc = reinterpret_cast<const uint4_nt*>(a.coef[c] + (static_cast<size_t>(img) * nblk + wave_block0) * 64u);
    uint4_nt rowv[8];
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t g = k * 64u + ln;
        rowv[k] = __builtin_nontemporal_load(wsrc + (g < wave_vecs ? g : wave_vecs - 1u));
    }
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t g = k * 64u + ln;
        stage[wv][(g >> 3) * 9u + (g & 7u)] = rowv[k];
    }
    wave_sync_lds();
    if (bidx >= nblk) return;
    // the luma scalers' islow runs with the FIRST component's table, as jpeg_idct_islow(cinfo, compptr, ...) does
    const uint32_t* q32 = reinterpret_cast<const uint32_t*>(a.qt + (static_cast<size_t>(img) * a.g.ncomp + c) * 64u);   // wave-uniform: scalar loads
    int32_t ws[8][8];
    int32_t dmax = 0, dmin = 0;                                 // extremes of the de-quantised coefficients (v_max3 / v_min3: one pair each)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (MODE == 1 && r == 4) {                              // jpeg_idct_4x4 never reads coefficient row 4
#pragma unroll
            for (int k = 0; k < 8; ++k) ws[r][k] = 0;
            continue;
        }
        const uint4_nt cv = stage[wv][ln * 9u + static_cast<uint32_t>(r)];
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t qq = q32[r * 4 + k];
            const int32_t d0 = __mul24(static_cast<int32_t>(static_cast<int16_t>(cw[k] & 0xffffu)), static_cast<int32_t>(qq & 0xffffu));   // 16-bit operands: exact
            const int32_t d1 = __mul24(static_cast<int32_t>(static_cast<int16_t>(cw[k] >> 16)), static_cast<int32_t>(qq >> 16));
            ws[r][2 * k] = d0; ws[r][2 * k + 1] = d1;
            dmax = max(dmax, max(d0, d1));
            dmin = min(dmin, min(d0, d1));
        }
    }
    const bool small = __all(dmax < (1 << 21) && dmin > -(1 << 21)) != 0;      // 24-bit multiplies in the first pass (see mulc)
    uint8_t* plane = a.plane[c] + static_cast<size_t>(img) * a.g.pw[c] * a.g.ph[c];
    // where the block goes is worked out behind the passes (the division's temporaries and results would otherwise sit in
    // registers next to the 64 coefficients and push some of them into scratch)
    uint32_t by = 0, bx = 0;
    auto locate = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        uint32_t b = bidx;
        asm volatile("" : "+v"(b));
        by = b / a.g.bw[c]; bx = b - by * a.g.bw[c];
    };

    if constexpr (MODE == 1) {
        locate();
        // jidctred.c jpeg_idct_4x4: columns 0-3, 5-7 (column 4 is not used by the second pass), rows 0-3
        int32_t col[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k == 4) continue;
            int32_t o[4];
            if (small) idct4_pass<true>(ws[0][k], ws[1][k], ws[2][k], ws[3][k], ws[5][k], ws[6][k], ws[7][k], o, 12);
            else idct4_pass<false>(ws[0][k], ws[1][k], ws[2][k], ws[3][k], ws[5][k], ws[6][k], ws[7][k], o, 12);
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r][k] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int32_t o[4];
            idct4_pass<true>(col[r][0], col[r][1], col[r][2], col[r][3], col[r][5], col[r][6], col[r][7], o, 19);
            const uint32_t v = range_limit(o[0]) | (range_limit(o[1]) << 8) | (range_limit(o[2]) << 16) | (range_limit(o[3]) << 24);
            *reinterpret_cast<uint32_t*>(plane + static_cast<size_t>(by * 4u + r) * a.g.pw[c] + bx * 4u) = v;
        }
        return;
    } else {
        // column pass (CONST_BITS - PASS1_BITS), in place; the multiply width is the wave's choice, made once
        if (small) bpl_column_pass<true>(ws);
        else bpl_column_pass<false>(ws);
        locate();
        if constexpr (MODE == 0) {
            // row pass (CONST_BITS + PASS1_BITS + 3: always in 24-bit range) + libjpeg's range-limit table: bits 18..27 of
            // a row-pass sum (rounding constant inside) are the table's index (v & 1023), the table is 1 KiB of LDS -- one
            // bit-field extract and one byte read per sample, no clamp, no special case for values beyond the clamp range
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int32_t in[8], out[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) in[k] = ws[r][k];
                idct8<true, true>(in, out, 18);
                uint32_t b[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) b[k] = lim8_lds[(static_cast<uint32_t>(out[k]) >> 18) & 1023u];
                const uint32_t w0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), w1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                *reinterpret_cast<uint2*>(plane + static_cast<size_t>(by * 8u + r) * a.g.pw[c] + bx * 8u) = make_uint2(w0, w1);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        } else {
            // row pass, then flow_scale_spatial[_srgb]_NxN (codecs_jpeg_idct_fast.c).  The sample never exists: bits 18..27
            // of a row-pass sum (rounding constant already inside) ARE the index (v & 1023) of libjpeg's range-limit table,
            // and the workgroup's table holds, under that index, the 12-bit linear light of the limited sample (_srgb
            // forms) or the limited sample itself -- one bit-field extract and one LDS read per sample, no clamp, no
            // special case for wrapped values.
            int32_t (&lin)[8][8] = ws;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int32_t in[8], out[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) in[k] = ws[r][k];
                idct8<true, true>(in, out, 18);
#pragma unroll
                for (int k = 0; k < 8; ++k) {    // 4-byte entries: the byte offset (sum >> 16) & 0xffc is one SDWA and (upper word, mask)
                    lin[r][k] = static_cast<int32_t>(*reinterpret_cast<const uint32_t*>(
                        reinterpret_cast<const uint8_t*>(lim_lds) + ((static_cast<uint32_t>(out[k]) >> 16) & 0xffcu)));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // rows combined with the integer weights of output row r, then columns with those of output column cc, rounded
            // by the two divisors' shift.  |weight| <= 117, linear <= 4095, sums of weights <= 512: every product fits
            // 24 x 24 -> 32 bits.
            if (srgb) bpl_scale_block<N, true>(lin, plane, by, bx, a.g.pw[c], l2s_lds);
            else bpl_scale_block<N, false>(lin, plane, by, bx, a.g.pw[c], l2s_lds);
        }
    }
}

// One launch per component (blockIdx.z: the two chroma components).  (One launch for all three with luma and chroma
// workgroups interleaved -- traffic-bound plain IDCTs beside the arithmetic-bound spatial scalers -- was measured: 410 us
// against 254 + 110 us for the 4/8 decode of 32 4K frames; every workgroup then carries the larger routine's registers.)
// The scaler's two tables: lim[i], i = (v & 1023) as libjpeg indexes its range-limit table, holds what the scaler reads of
// the limited sample -- its 12-bit linear light (_srgb forms) or the sample; l2s is lut_linear_to_srgb.
__device__ __forceinline__ void bpl_tables(const JpegArgs& a, bool srgb, uint32_t* lim_lds, uint8_t* l2s_lds) {
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) {
        const uint32_t sample = range_limit(static_cast<int32_t>(i));
        lim_lds[i] = srgb ? static_cast<uint32_t>(a.sc.s2l[sample]) : sample;
    }
    if (srgb && threadIdx.x < 256u) reinterpret_cast<uint4*>(l2s_lds)[threadIdx.x] = reinterpret_cast<const uint4*>(a.sc.l2s)[threadIdx.x];
    __syncthreads();
}

template <int MODE, int N = 1>       // N: block size of the spatial scaler (MODE 2)
__global__ void __launch_bounds__(bpl_threads(MODE)) __attribute__((amdgpu_waves_per_eu(4, 4)))     // 4 waves per SIMD: <= 128 registers
jpeg_idct_block_per_lane_kernel(const JpegArgs a) {
    // one block, tables first: their LDS addresses fit the 16-bit offset field of the table reads
    struct Lds {                                                // MODE 2: 4 + 4 + 72 KiB = exactly half of a CU's 160 KiB
        union {
            uint32_t lim[MODE == 2 ? 1024 : 4];
            uint8_t lim8[MODE == 0 ? 1024 : 16];
        };
        uint8_t l2s[MODE == 2 ? 4096 : 16];
        uint4_nt stage[bpl_threads(MODE) / 64][64 * 9];
    };
    static_assert(sizeof(Lds) <= (MODE == 2 ? 80 : 40) * 1024, "two (scaler forms) / four (plain forms) workgroups per CU");
    __shared__ __attribute__((aligned(16))) Lds lds;
    if (MODE == 2) bpl_tables(a, a.g.luma_mode == 2u, lds.lim, lds.l2s);
    if (MODE == 0) {
        for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) lds.lim8[i] = static_cast<uint8_t>(range_limit(static_cast<int32_t>(i)));
        __syncthreads();
    }
    idct_block_per_lane<MODE, N>(a, a.comp + blockIdx.z, blockIdx.y, blockIdx.x, lds.stage, lds.lim, lds.l2s, lds.lim8);
}

// ---- up-sample + colour ------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t chroma_at(const uint8_t* p, uint32_t pw, uint32_t dw, uint32_t dh, int32_t x, int32_t y) {
    x = x < 0 ? 0 : (x >= static_cast<int32_t>(dw) ? static_cast<int32_t>(dw) - 1 : x);   // edge duplication (jdmainct.c)
    y = y < 0 ? 0 : (y >= static_cast<int32_t>(dh) ? static_cast<int32_t>(dh) - 1 : y);
    return p[static_cast<size_t>(y) * pw + static_cast<size_t>(x)];
}

__device__ __forceinline__ uint32_t clamp255(int32_t v) { return static_cast<uint32_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__device__ __forceinline__ uint32_t ycc_to_bgra(int32_t Y, int32_t cbv, int32_t crv) {   // jdcolor.c tables, in place
    const int32_t cb = cbv - 128, cr = crv - 128;                     // |cb|, |cr| <= 128: 24-bit multiplies are exact
    const int32_t r = Y + ((__mul24(91881, cr) + 32768) >> 16);
    const int32_t g = Y + ((__mul24(-22554, cb) + 32768 + __mul24(-46802, cr)) >> 16);
    const int32_t b = Y + ((__mul24(116130, cb) + 32768) >> 16);
    return clamp255(b) | (clamp255(g) << 8) | (clamp255(r) << 16) | 0xff000000u;
}

// 4 neighbouring chroma samples of one row starting at column c0 (may be -1), edge-duplicated like libjpeg:
// interior lanes use one unaligned 4-byte load, lanes at the left/right edge fall back to clamped byte loads.
__device__ __forceinline__ void chroma4(const uint8_t* P, uint32_t W, uint32_t DW, uint32_t DH, int32_t c0, int32_t y,
                                        int32_t (&o)[4]) {
    y = y < 0 ? 0 : (y >= static_cast<int32_t>(DH) ? static_cast<int32_t>(DH) - 1 : y);
    const uint8_t* row = P + static_cast<size_t>(y) * W;
    if (c0 >= 0 && c0 + 3 < static_cast<int32_t>(DW)) {
        uint32_t v;
        __builtin_memcpy(&v, row + c0, 4);
        o[0] = v & 255u; o[1] = (v >> 8) & 255u; o[2] = (v >> 16) & 255u; o[3] = v >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int32_t x = c0 + k;
            x = x < 0 ? 0 : (x >= static_cast<int32_t>(DW) ? static_cast<int32_t>(DW) - 1 : x);
            o[k] = row[x];
        }
    }
}

// the same four samples, packed into one word (byte k = sample k): lets the colour pass request every chroma row of its
// 16-row tile up front and unpack at use
__device__ __forceinline__ uint32_t chroma4_packed(const uint8_t* P, uint32_t W, uint32_t DW, uint32_t DH, int32_t c0, int32_t y) {
    y = y < 0 ? 0 : (y >= static_cast<int32_t>(DH) ? static_cast<int32_t>(DH) - 1 : y);
    const uint8_t* row = P + static_cast<size_t>(y) * W;
    if (c0 >= 0 && c0 + 3 < static_cast<int32_t>(DW)) {
        uint32_t v;
        __builtin_memcpy(&v, row + c0, 4);
        return v;
    }
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int32_t x = c0 + k;
        x = x < 0 ? 0 : (x >= static_cast<int32_t>(DW) ? static_cast<int32_t>(DW) - 1 : x);
        v |= static_cast<uint32_t>(row[x]) << (8 * k);
    }
    return v;
}
__device__ __forceinline__ void unpack4(uint32_t v, int32_t (&o)[4]) {
    o[0] = v & 255u; o[1] = (v >> 8) & 255u; o[2] = (v >> 16) & 255u; o[3] = v >> 24;
}

// One lane = 4 horizontally adjacent output pixels x kColorRows output rows (one 4-byte Y load and one 16-byte BGRA store
// per row).  With h2v2 fancy up-sampling a row pair (2cy, 2cy+1) needs chroma rows cy-1, cy, cy+1; walking down the rows
// the lane keeps the last two chroma rows in registers and loads one new row per pair, so a pixel costs 0.25 Y loads
// + 0.25 chroma loads instead of 1.25, and a wave lives 16 rows instead of one.
constexpr uint32_t kColorRows = 16;

constexpr uint32_t kTilePx = 1024, kYsPitch = kTilePx + 16;      // luma tile of the fused form: 16 rows x 1024 pixels


// FUSED_LUMA (full-size decode of 3-component files): the workgroup first runs the islow IDCT of the 2 x 128 luma
// blocks under its 1024 x 16 pixel tile into LDS -- 8 passes of 32 blocks, 8 lanes per block as in jpeg_idct_kernel, ordered
// by wave-scope fences -- and the colour pass reads Y from there: the luma plane never exists in HBM (1 B/px written
// + 1 B/px read less, one launch less).
template <bool FUSED_LUMA>
__global__ void __launch_bounds__(256) jpeg_color_kernel(const JpegArgs a) {
    __shared__ int32_t ws[FUSED_LUMA ? kBlocksPerWg * kBlockPitch : 1];
    __shared__ __attribute__((aligned(16))) uint8_t ys[FUSED_LUMA ? kColorRows * kYsPitch : 16];
    const uint32_t img = blockIdx.z;
    if (FUSED_LUMA) {
        const uint32_t t = threadIdx.x, lane8 = t & 7u, lb = t >> 3;
        int32_t* w = ws + lb * kBlockPitch;
        const uint32_t nblk = a.g.bw[0] * a.g.bh[0];
        const uint16_t* q = a.qt + static_cast<size_t>(img) * a.g.ncomp * 64u + lane8 * 8u;
        const uint4 qv = *reinterpret_cast<const uint4*>(q);
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
        // All eight passes' coefficient rows are requested before the first pass runs (8 x 16 B per lane in flight): one
        // load per pass, waited for on the spot, left a workgroup with 8 dependent HBM round trips and the CU with a few
        // KB outstanding -- the kernel was bound by that latency, not by bandwidth or arithmetic.
        uint4_nt cvs[8];
#pragma unroll
        for (uint32_t pass = 0; pass < 8u; ++pass) {
            const uint32_t bl = pass * kBlocksPerWg + lb;
            const uint32_t by = min(blockIdx.y * 2u + (bl >> 7), a.g.bh[0] - 1u), bx = min(blockIdx.x * 128u + (bl & 127u), a.g.bw[0] - 1u);
            const int16_t* src = a.coef[0] + (static_cast<size_t>(img) * nblk + static_cast<size_t>(by) * a.g.bw[0] + bx) * 64u + lane8 * 8u;
            cvs[pass] = __builtin_nontemporal_load(reinterpret_cast<const uint4_nt*>(src));
        }
#pragma unroll
        for (uint32_t pass = 0; pass < 8u; ++pass) {
            const uint32_t bl = pass * kBlocksPerWg + lb;                 // 0..255: block row bl >> 7, block column bl & 127
            const uint32_t by = blockIdx.y * 2u + (bl >> 7), bx = blockIdx.x * 128u + (bl & 127u);
            const bool on = bx < a.g.bw[0] && by < a.g.bh[0];
            if (on) {
                const uint4_nt cv = cvs[pass];
                const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t c0 = static_cast<int16_t>(cw[k] & 0xffffu), c1 = static_cast<int16_t>(cw[k] >> 16);
                    w[lane8 * 8u + 2 * k] = __mul24(c0, static_cast<int32_t>(qw[k] & 0xffffu));
                    w[lane8 * 8u + 2 * k + 1] = __mul24(c1, static_cast<int32_t>(qw[k] >> 16));
                }
            }
            wave_sync_lds();
            int32_t in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (on) {
#pragma unroll
                for (int r = 0; r < 8; ++r) in[r] = w[r * 8 + lane8];
            }
            const bool small = wave_all_small(in, on);
            if (on) {                                                   // column pass: lane8 = column
                int32_t out[8];
                idct8_pass1(in, out, small);
                wave_sync_lds();
#pragma unroll
                for (int r = 0; r < 8; ++r) w[r * 8 + lane8] = out[r];
            }
            wave_sync_lds();
            if (on) {                                                   // row pass: lane8 = row
                int32_t in[8], out[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) in[k] = w[lane8 * 8u + k];
                idct8<true>(in, out, 18);
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo |= range_limit(out[k]) << (8 * k);
                    hi |= range_limit(out[4 + k]) << (8 * k);
                }
                *reinterpret_cast<uint2*>(&ys[((bl >> 7) * 8u + lane8) * kYsPitch + (bl & 127u) * 8u]) = make_uint2(lo, hi);
            }
            wave_sync_lds();
        }
        __syncthreads();
    }
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (x0 >= a.g.out_w) return;
    const uint32_t y_begin = blockIdx.y * kColorRows;
    const uint32_t y_end = min(y_begin + kColorRows, a.g.out_h);
    const uint8_t* py = a.plane[0] + static_cast<size_t>(img) * a.g.pw[0] * a.g.ph[0] + x0;
    uint8_t* dst0 = a.bgra + static_cast<size_t>(img) * a.image_bytes + static_cast<size_t>(x0) * 4u;
    const bool vec_store = x0 + 4u <= a.g.out_w && ((reinterpret_cast<uintptr_t>(dst0) | a.stride) & 15u) == 0u;

    auto emit = [&](uint32_t y, const int32_t (&v)[2][4], bool gray) {
        uint32_t yv;
        if (FUSED_LUMA) yv = *reinterpret_cast<const uint32_t*>(&ys[(y - y_begin) * kYsPitch + threadIdx.x * 4u]);
        else __builtin_memcpy(&yv, py + static_cast<size_t>(y) * a.g.pw[0], 4);     // planes carry 16 bytes of slack behind the last row
        const int32_t Y[4] = {static_cast<int32_t>(yv & 255u), static_cast<int32_t>((yv >> 8) & 255u),
                              static_cast<int32_t>((yv >> 16) & 255u), static_cast<int32_t>(yv >> 24)};
        uint32_t out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            out[i] = gray ? (static_cast<uint32_t>(Y[i]) * 0x010101u | 0xff000000u) : ycc_to_bgra(Y[i], v[0][i], v[1][i]);
        uint8_t* dst = dst0 + static_cast<size_t>(y) * a.stride;
        if (vec_store) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i)
                if (x0 + i < a.g.out_w) reinterpret_cast<uint32_t*>(dst)[i] = out[i];
        }
    };

    int32_t v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (a.g.ncomp == 1) {
        for (uint32_t y = y_begin; y < y_end; ++y) emit(y, v, true);
        return;
    }
    const uint8_t* P[2] = {a.plane[1] + static_cast<size_t>(img) * a.g.pw[1] * a.g.ph[1],
                           a.plane[2] + static_cast<size_t>(img) * a.g.pw[2] * a.g.ph[2]};
    if (a.g.upsample == 0u) {                                         // planes already at output resolution
        for (uint32_t y = y_begin; y < y_end; ++y) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint32_t cv;
                __builtin_memcpy(&cv, P[k] + static_cast<size_t>(y) * a.g.pw[1 + k] + x0, 4);
                v[k][0] = cv & 255u; v[k][1] = (cv >> 8) & 255u; v[k][2] = (cv >> 16) & 255u; v[k][3] = cv >> 24;
            }
            emit(y, v, false);
        }
        return;
    }
    // columns cx-1 .. cx+2 around the two chroma samples (cx = x0/2, cx+1) under this lane's 4 pixels
    const int32_t c0 = static_cast<int32_t>(x0 >> 1) - 1;
    if (a.g.upsample == 1u) {                                         // h2v1 fancy: (3*near + far + {1,2}) >> 2
        for (uint32_t y = y_begin; y < y_end; ++y) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int32_t s[4];
                chroma4(P[k], a.g.pw[1 + k], a.g.dw[1 + k], a.g.dh[1 + k], c0, static_cast<int32_t>(y), s);
                v[k][0] = (3 * s[1] + s[0] + 1) >> 2; v[k][1] = (3 * s[1] + s[2] + 2) >> 2;
                v[k][2] = (3 * s[2] + s[1] + 1) >> 2; v[k][3] = (3 * s[2] + s[3] + 2) >> 2;
            }
            emit(y, v, false);
        }
        return;
    }
    if (a.g.upsample >= 3u) {              // rare forms (parity paths): replication, and the vertical-only triangle of 4:4:0
        for (uint32_t y = y_begin; y < y_end; ++y) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t W = a.g.pw[1 + k], DW = a.g.dw[1 + k], DH = a.g.dh[1 + k];
                if (a.g.upsample == 3u) {                              // h2v1, samples replicated
                    int32_t s[4];
                    chroma4(P[k], W, DW, DH, c0, static_cast<int32_t>(y), s);
                    v[k][0] = s[1]; v[k][1] = s[1]; v[k][2] = s[2]; v[k][3] = s[2];
                } else if (a.g.upsample == 4u) {                       // h2v2, samples replicated
                    int32_t s[4];
                    chroma4(P[k], W, DW, DH, c0, static_cast<int32_t>(y >> 1), s);
                    v[k][0] = s[1]; v[k][1] = s[1]; v[k][2] = s[2]; v[k][3] = s[2];
                } else {                                               // h1v2: 5 fancy (3*near + far + {1,2}) >> 2, 6 replicated
                    const int32_t cy = static_cast<int32_t>(y >> 1);
                    int32_t near[4], far[4];
                    chroma4(P[k], W, DW, DH, static_cast<int32_t>(x0), cy, near);
                    if (a.g.upsample == 6u) { for (int i = 0; i < 4; ++i) v[k][i] = near[i]; }
                    else {
                        chroma4(P[k], W, DW, DH, static_cast<int32_t>(x0), (y & 1u) ? cy + 1 : cy - 1, far);
                        const int32_t bias = (y & 1u) ? 2 : 1;
                        for (int i = 0; i < 4; ++i) v[k][i] = (3 * near[i] + far[i] + bias) >> 2;
                    }
                }
            }
            emit(y, v, false);
        }
        return;
    }
    // h2v2 fancy: triangle in both directions; y_begin is even (kColorRows is), rows come in pairs (2cy, 2cy+1).
    // The tile's chroma rows cy0 - 1 .. cy0 + kColorRows / 2 are requested up front (one packed word each).
    constexpr int kCRows = kColorRows / 2 + 2;
    uint32_t craw[2][kCRows];
    const int32_t cy0 = static_cast<int32_t>(y_begin >> 1);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < kCRows; ++r)
            craw[k][r] = chroma4_packed(P[k], a.g.pw[1 + k], a.g.dw[1 + k], a.g.dh[1 + k], c0, cy0 - 1 + r);
    int32_t prev[2][4], cur[2][4], next[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) { unpack4(craw[k][0], prev[k]); unpack4(craw[k][1], cur[k]); }
#pragma unroll
    for (uint32_t pr = 0; pr < kColorRows / 2u; ++pr) {
        const uint32_t y = y_begin + 2u * pr;
        if (y >= y_end) break;
#pragma unroll
        for (int k = 0; k < 2; ++k) unpack4(craw[k][pr + 2u], next[k]);
#pragma unroll
        for (uint32_t r = 0; r < 2u; ++r) {
            if (y + r >= y_end) break;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int32_t s[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = 3 * cur[k][i] + (r ? next[k][i] : prev[k][i]);
                v[k][0] = (3 * s[1] + s[0] + 8) >> 4; v[k][1] = (3 * s[1] + s[2] + 7) >> 4;
                v[k][2] = (3 * s[2] + s[1] + 8) >> 4; v[k][3] = (3 * s[2] + s[3] + 7) >> 4;
            }
            emit(y + r, v, false);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) { prev[k][i] = cur[k][i]; cur[k][i] = next[k][i]; }
    }
}

}  // namespace ifhip

// ==================================================================================================================
using namespace ifhip;
#include "block_scalers.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace {
struct ScalerDeviceTables { uint16_t* s2l = nullptr; uint8_t* l2s = nullptr; };
std::mutex g_sc_mu;
std::map<int, ScalerDeviceTables> g_sc_tables;

int scaler_device_tables(ScalerDeviceTables* out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device; this library has no CPU path");
    const BlockScalerTables* t = block_scaler_tables();
    if (!t) return fail(IFHIP_INVALID_STATE, "InvalidState: block scaler tables could not be generated");
    std::lock_guard<std::mutex> lk(g_sc_mu);
    auto it = g_sc_tables.find(dev);
    if (it == g_sc_tables.end()) {
        ScalerDeviceTables d;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.s2l), sizeof t->srgb_to_linear));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.l2s), sizeof t->linear_to_srgb));
        HIP_TRY(hipMemcpy(d.s2l, t->srgb_to_linear, sizeof t->srgb_to_linear, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.l2s, t->linear_to_srgb, sizeof t->linear_to_srgb, hipMemcpyHostToDevice));
        it = g_sc_tables.emplace(dev, d).first;
    }
    *out = it->second;
    return IFHIP_OK;
}
}  // namespace


struct ifhip_jpeg_stage {
    int device = -1;
    JpegGeom g;
    uint32_t max_images = 0;
    uint8_t* planes[3] = {nullptr, nullptr, nullptr};
    ~ifhip_jpeg_stage() { for (auto* p : planes) if (p) (void)DEV_FREE(p); }
};


// jdmaster.c: a component's IDCT size starts at scale_num and doubles while it stays below 8 before doubling and both
// sampling ratios allow it (4:2:0 chroma decodes at 2 * scale_num, 4:2:2 / 4:4:0 chroma stays at scale_num)
static uint32_t component_idct_size(uint32_t scale_num, uint32_t hs_c, uint32_t vs_c, uint32_t hmax, uint32_t vmax) {
    uint32_t ssize = scale_num;
    while (ssize < 8u && (hmax * scale_num) % (hs_c * ssize * 2u) == 0u && (vmax * scale_num) % (vs_c * ssize * 2u) == 0u) ssize *= 2u;
    return ssize;
}

static int make_geom(uint32_t width, uint32_t height, int ncomp, const uint8_t* hs, const uint8_t* vs, int scale_num,
                     int luma_spatial, int luma_srgb, JpegGeom* g) {
    if (scale_num < 1 || scale_num > 8 || scale_num == 7)            // 7/8 is never asked for (mozjpeg_decoder.rs:603-606)
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: jpeg scale_num %d/8 (supported: 1..6, 8)", scale_num);
    if (width == 0 || height == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Bitmap dimensions cannot be zero");
    if (ncomp != 1 && ncomp != 3) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-component JPEG", ncomp);
    std::memset(g, 0, sizeof *g);
    g->width = width; g->height = height; g->ncomp = ncomp; g->hmax = g->vmax = 1;
    for (int c = 0; c < ncomp; ++c) {
        g->hs[c] = (ncomp == 1) ? 1u : hs[c]; g->vs[c] = (ncomp == 1) ? 1u : vs[c];
        if (g->hs[c] < 1 || g->hs[c] > 2 || g->vs[c] < 1 || g->vs[c] > 2)
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: sampling factor %ux%u", g->hs[c], g->vs[c]);
        g->hmax = g->hs[c] > g->hmax ? g->hs[c] : g->hmax;
        g->vmax = g->vs[c] > g->vmax ? g->vs[c] : g->vmax;
    }
    if (ncomp == 3) {
        const bool chroma_1x1 = g->hs[1] == 1 && g->vs[1] == 1 && g->hs[2] == 1 && g->vs[2] == 1;
        const bool luma_max = g->hs[0] == g->hmax && g->vs[0] == g->vmax;
        if (!chroma_1x1 || !luma_max)
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: only 4:4:4, 4:2:2 (h2v1), 4:4:0 (h1v2) and 4:2:0 sampling");
    }
    g->scale_num = static_cast<uint32_t>(scale_num);
    g->luma_mode = (scale_num < 8 && luma_spatial) ? (luma_srgb ? 2u : 1u) : 0u;     // codec_jpeg_wrapper.c:285-339
    g->out_w = (width * g->scale_num + 7u) / 8u;                                      // jpeg_calc_output_dimensions
    g->out_h = (height * g->scale_num + 7u) / 8u;
    const uint32_t mw = (width + 8u * g->hmax - 1u) / (8u * g->hmax), mh = (height + 8u * g->vmax - 1u) / (8u * g->vmax);
    g->blocks_before[0] = 0;
    for (int c = 0; c < ncomp; ++c) {
        g->bw[c] = mw * g->hs[c]; g->bh[c] = mh * g->vs[c];
        const uint32_t n = component_idct_size(g->scale_num, g->hs[c], g->vs[c], g->hmax, g->vmax);
        g->idct_n[c] = n;
        g->pw[c] = g->bw[c] * n; g->ph[c] = g->bh[c] * n;
        g->dw[c] = (width * g->hs[c] * n + g->hmax * 8u - 1u) / (g->hmax * 8u);         // downsampled_width at this IDCT size
        g->dh[c] = (height * g->vs[c] * n + g->vmax * 8u - 1u) / (g->vmax * 8u);
        g->blocks_before[c + 1] = g->blocks_before[c] + g->bw[c] * g->bh[c];
    }
    // jdsample.c jinit_upsampler: what is left to up-sample, and whether the triangle ("fancy") forms apply:
    // do_fancy_upsampling (libjpeg's default; the reference never touches it) needs min_DCT_scaled_size > 1, and the
    // h2v1 / h2v2 forms need downsampled_width > 2 -- otherwise samples are replicated.
    g->upsample = 0u;
    if (ncomp == 3) {
        const uint32_t ux = (g->hmax * g->scale_num) / (g->hs[1] * g->idct_n[1]), uy = (g->vmax * g->scale_num) / (g->vs[1] * g->idct_n[1]);
        const bool fancy = g->scale_num > 1u;
        if (ux == 2u && uy == 1u) g->upsample = (fancy && g->dw[1] > 2u) ? 1u : 3u;
        else if (ux == 2u && uy == 2u) g->upsample = (fancy && g->dw[1] > 2u) ? 2u : 4u;
        else if (ux == 1u && uy == 2u) g->upsample = fancy ? 5u : 6u;
        else if (ux != 1u || uy != 1u) return fail(IFHIP_INVALID_STATE, "InvalidState: up-sampling %ux%u", ux, uy);
    }
    return IFHIP_OK;
}

extern "C" {

int ifhip_jpeg_stage_create(ifhip_jpeg_stage** stage, uint32_t width, uint32_t height, int n_components,
                            const uint8_t* h_samp, const uint8_t* v_samp, int scale_num, int luma_spatial,
                            int luma_srgb, uint32_t max_images) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage out-pointer");
    *stage = nullptr;
    if (max_images == 0 || (n_components == 3 && (!h_samp || !v_samp)))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: jpeg stage needs sampling factors and max_images >= 1");
    std::unique_ptr<ifhip_jpeg_stage> s(new ifhip_jpeg_stage);
    int rc = make_geom(width, height, n_components, h_samp, v_samp, scale_num, luma_spatial, luma_srgb, &s->g);
    if (rc) return rc;
    if (s->g.height > 65535u || max_images > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: more than 65535 rows/images per launch");
    if (int arc = require_gfx950(&s->device)) return arc;
    s->max_images = max_images;
    for (int c = 0; c < n_components; ++c)
        HIP_TRY(DEV_MALLOC(&s->planes[c], static_cast<size_t>(s->g.pw[c]) * s->g.ph[c] * max_images + 16));
    *stage = s.release();
    return IFHIP_OK;
}

void ifhip_jpeg_stage_destroy(ifhip_jpeg_stage* stage) { delete stage; }

int ifhip_jpeg_stage_output_size(const ifhip_jpeg_stage* stage, uint32_t* out_w, uint32_t* out_h) {
    if (!stage || !out_w || !out_h) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    *out_w = stage->g.out_w; *out_h = stage->g.out_h;
    return IFHIP_OK;
}

int ifhip_jpeg_stage_block_dims(const ifhip_jpeg_stage* stage, uint32_t* blocks_w3, uint32_t* blocks_h3) {
    if (!stage || !blocks_w3 || !blocks_h3) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    for (int c = 0; c < 3; ++c) { blocks_w3[c] = c < stage->g.ncomp ? stage->g.bw[c] : 0; blocks_h3[c] = c < stage->g.ncomp ? stage->g.bh[c] : 0; }
    return IFHIP_OK;
}

// Arguments of a stage call, validated; the launches of the component IDCTs (planes in HBM).
static int stage_args(ifhip_jpeg_stage* stage, const int16_t* d_coef0, const int16_t* d_coef1, const int16_t* d_coef2,
                      const uint16_t* d_qt, uint32_t n_images, JpegArgs* out) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage");
    if (n_images > stage->max_images) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: %u images exceed the stage capacity %u", n_images, stage->max_images);
    if (!d_coef0 || !d_qt || (stage->g.ncomp == 3 && (!d_coef1 || !d_coef2)))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null coefficient / table / bitmap pointer");
    if ((reinterpret_cast<uintptr_t>(d_coef0) | reinterpret_cast<uintptr_t>(d_coef1) | reinterpret_cast<uintptr_t>(d_coef2)
         | reinterpret_cast<uintptr_t>(d_qt)) & 15u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient planes and quantisation tables must be 16-byte aligned");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != stage->device) return fail(IFHIP_INVALID_STATE, "InvalidState: stage belongs to device %d, current device is %d", stage->device, dev);
    JpegArgs& a = *out;
    std::memset(&a, 0, sizeof a);
    a.g = stage->g;
    a.coef[0] = d_coef0; a.coef[1] = d_coef1; a.coef[2] = d_coef2;
    a.qt = d_qt;
    for (int c = 0; c < 3; ++c) a.plane[c] = stage->planes[c];
    a.n_images = n_images;
    if (a.g.luma_mode != 0u) {
        ScalerDeviceTables dt;
        int rc = scaler_device_tables(&dt);
        if (rc) return rc;
        const BlockScalerTables* t = block_scaler_tables();
        const uint32_t n = a.g.idct_n[0];
        for (int i = 0; i < 7; ++i) {
            a.sc.log2_div[i] = t->scaler[n].log2_div[i];
            for (int j = 0; j < 8; ++j) a.sc.w[i][j] = t->scaler[n].w[i][j];
        }
        a.sc.s2l = dt.s2l; a.sc.l2s = dt.l2s;
    }
    const uint64_t total_blocks = static_cast<uint64_t>(a.g.blocks_before[a.g.ncomp]) * n_images;
    if ((total_blocks + kBlocksPerWg - 1) / kBlocksPerWg > 0x7fffffffull) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: batch too large for one launch");
    return IFHIP_OK;
}

// block-per-lane routine of component c: 0 islow 8x8, 1 jidctred 4x4, 2 islow + spatial scaler; -1: the eight-lanes kernel
static int bpl_mode(const JpegGeom& g, int c) {
    const uint32_t n = g.idct_n[c];
    if (c == 0 && g.luma_mode != 0u && n < 8u) return 2;
    return n == 8u ? 0 : n == 4u ? 1 : -1;
}

static int launch_idct_planes(JpegArgs a, int first_component, hipStream_t st) {
    const JpegGeom& g = a.g;
    for (int c = first_component; c < g.ncomp; ++c) {
        a.comp = static_cast<uint32_t>(c);
        const uint32_t nblk = g.bw[c] * g.bh[c];
        // the two chroma components have the same geometry: one launch covers both (blockIdx.z)
        const bool both = c == 1 && g.ncomp == 3 && g.bw[1] == g.bw[2] && g.bh[1] == g.bh[2] && g.idct_n[1] == g.idct_n[2];
        // the common block routines run one lane per block; the rarer sizes keep the eight-lanes-per-block kernel
        const int bpl = bpl_mode(g, c);
        const uint32_t bt = bpl_threads(bpl);
        const dim3 bgrid((nblk + bt - 1u) / bt, a.n_images, both ? 2u : 1u);
        if (bpl == 0) hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<0>), bgrid, dim3(bt), 0, st, a);
        else if (bpl == 1) hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<1>), bgrid, dim3(bt), 0, st, a);
        else if (bpl == 2) {
            switch (g.idct_n[0]) {                              // the scaler's weights are compile-time data of each instantiation
            case 1: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 1>), bgrid, dim3(bt), 0, st, a); break;
            case 2: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 2>), bgrid, dim3(bt), 0, st, a); break;
            case 3: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 3>), bgrid, dim3(bt), 0, st, a); break;
            case 4: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 4>), bgrid, dim3(bt), 0, st, a); break;
            case 5: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 5>), bgrid, dim3(bt), 0, st, a); break;
            case 6: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 6>), bgrid, dim3(bt), 0, st, a); break;
            default: hipLaunchKernelGGL((jpeg_idct_block_per_lane_kernel<2, 7>), bgrid, dim3(bt), 0, st, a); break;
            }
        }
        else hipLaunchKernelGGL(jpeg_idct_kernel, dim3((nblk + kBlocksPerWg - 1) / kBlocksPerWg, a.n_images, both ? 2u : 1u), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
        if (both) ++c;
    }
    return IFHIP_OK;
}

int ifhip_jpeg_idct_color_batch_device(ifhip_jpeg_stage* stage, const int16_t* d_coef0, const int16_t* d_coef1,
                                       const int16_t* d_coef2, const uint16_t* d_qt, uint32_t n_images,
                                       uint8_t* d_bgra, size_t image_bytes, uint32_t stride, void* hip_stream) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage");
    if (n_images == 0) return IFHIP_OK;
    if (!d_bgra) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null coefficient / table / bitmap pointer");
    if (static_cast<uint64_t>(stage->g.out_w) * 4u > stride || (stride & 3u) || (image_bytes & 3u) || (reinterpret_cast<uintptr_t>(d_bgra) & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: BGRA rows must be 4-byte aligned and stride >= 4*w");
    JpegArgs a;
    int rc = stage_args(stage, d_coef0, d_coef1, d_coef2, d_qt, n_images, &a);
    if (rc) return rc;
    a.bgra = d_bgra; a.image_bytes = image_bytes; a.stride = stride;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // full-size colour decode: the luma IDCT runs inside the colour kernel (no luma plane in HBM)
    const bool fused_luma = a.g.ncomp == 3 && a.g.scale_num == 8u && a.g.luma_mode == 0u;
    rc = launch_idct_planes(a, fused_luma ? 1 : 0, st);
    if (rc) return rc;
    const dim3 cgrid((a.g.out_w + 1023u) / 1024u, (a.g.out_h + kColorRows - 1u) / kColorRows, n_images);
    if (fused_luma) hipLaunchKernelGGL((jpeg_color_kernel<true>), cgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((jpeg_color_kernel<false>), cgrid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

// MozJpegDecoder::read_frame (mozjpeg_decoder.rs:346-362) feeding DrawImageDef::render (scale_render.rs:304-313) as ONE
// device call: coefficient planes in, the resampled canvas out.  When the component planes come out of the IDCT at output
// resolution (every 4:2:0 file decoded at 1/8 .. 4/8 -- chroma then runs the twice-larger IDCT -- and 4:4:4 at any scale),
// the resampler reads them directly and converts YCbCr -> RGB in its row fetch: the decoded BGRA frame never exists in
// HBM (*fused = 1).  Everything else (fancy up-sampling, grayscale, shapes the fused resampler does not take) goes through a
// stream-ordered BGRA scratch and the two calls this replaces (*fused = 0): same bytes either way.
int ifhip_jpeg_decode_resample_batch_device(ifhip_jpeg_stage* stage, const int16_t* d_coef0, const int16_t* d_coef1,
                                            const int16_t* d_coef2, const uint16_t* d_qt, uint32_t n_images,
                                            const ifhip_resample_plan* plan, uint8_t* d_canvas, size_t canvas_image_bytes,
                                            uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride, uint32_t x, uint32_t y,
                                            int working_space, int compositing, uint32_t matte_bgra, int* fused, void* hip_stream) {
    if (fused) *fused = 0;
    if (!stage || !plan) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage / plan");
    uint32_t pin_w = 0, pin_h = 0, pout_w = 0, pout_h = 0;
    resample_plan_shape(plan, &pin_w, &pin_h, &pout_w, &pout_h);
    if (pin_w != stage->g.out_w || pin_h != stage->g.out_h)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: the plan resamples %ux%u frames, the stage decodes to %ux%u", pin_w, pin_h, stage->g.out_w, stage->g.out_h);
    if (n_images == 0) return IFHIP_OK;
    JpegArgs a;
    int rc = stage_args(stage, d_coef0, d_coef1, d_coef2, d_qt, n_images, &a);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const JpegGeom& g = a.g;
    const bool planes_at_output = g.ncomp == 3 && g.upsample == 0u && g.pw[0] == g.pw[1] && g.pw[0] == g.pw[2] &&
                                  g.ph[0] == g.ph[1] && g.ph[0] == g.ph[2];
    // the resampler is asked FIRST whether it takes the planes (alignment, a shape of the planar-source instantiations):
    // a refusal behind the plane IDCTs would run every IDCT twice
    if (planes_at_output &&
        resample_from_ycc_planes_v(plan, a.plane[0], a.plane[1], a.plane[2], static_cast<size_t>(g.pw[0]) * g.ph[0], g.pw[0], n_images,
                                   d_canvas, canvas_image_bytes, canvas_w, canvas_h, canvas_stride, x, y, working_space, compositing,
                                   matte_bgra, hip_stream, true) == IFHIP_OK) {
        rc = launch_idct_planes(a, 0, st);
        if (rc) return rc;
        rc = resample_from_ycc_planes_v(plan, a.plane[0], a.plane[1], a.plane[2], static_cast<size_t>(g.pw[0]) * g.ph[0], g.pw[0], n_images,
                                      d_canvas, canvas_image_bytes, canvas_w, canvas_h, canvas_stride, x, y, working_space, compositing,
                                      matte_bgra, hip_stream);
        if (rc != kNotFusable) {
            if (rc == IFHIP_OK && fused) *fused = 1;
            return rc;
        }
    }
    const uint32_t stride = ifhip_stride_for_width(g.out_w);
    const size_t image_bytes = static_cast<size_t>(stride) * g.out_h;
    uint8_t* scratch = nullptr;
    HIP_TRY(static_cast<hipError_t>(cached_malloc_for_stream(reinterpret_cast<void**>(&scratch), image_bytes * n_images, st, true)));
    rc = ifhip_jpeg_idct_color_batch_device(stage, d_coef0, d_coef1, d_coef2, d_qt, n_images, scratch, image_bytes, stride, hip_stream);
    if (rc == IFHIP_OK)
        rc = ifhip_scale_and_render_batch_device(plan, scratch, image_bytes, stride, 0, n_images, d_canvas, canvas_image_bytes, canvas_w,
                                                 canvas_h, canvas_stride, x, y, working_space, compositing, matte_bgra, nullptr, -1, hip_stream);
    const hipError_t fe = static_cast<hipError_t>(cached_free_after(scratch, st));
    if (rc) return rc;
    HIP_TRY(fe);
    return IFHIP_OK;
}

int ifhip_jpeg_idct_color(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                          int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                          uint32_t height, int scale_num, int luma_spatial, int luma_srgb, uint8_t* bgra, uint32_t stride) {
    JpegGeom g;
    int rc = make_geom(width, height, n_components, h_samp, v_samp, scale_num, luma_spatial, luma_srgb, &g);
    if (rc) return rc;
    if (!coef0 || !qt || !bgra || (n_components == 3 && (!coef1 || !coef2)))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null coefficient / table / bitmap pointer");
    if (static_cast<uint64_t>(g.out_w) * 4u > stride || (stride & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: stride smaller than a BGRA row or not a multiple of 4");
    ifhip_jpeg_stage* stage = nullptr;
    rc = ifhip_jpeg_stage_create(&stage, width, height, n_components, h_samp, v_samp, scale_num, luma_spatial, luma_srgb, 1);
    width = g.out_w; height = g.out_h;          // the bitmap the caller handed in has the scaled size
    if (rc) return rc;
    std::unique_ptr<ifhip_jpeg_stage> guard(stage);
    const int16_t* hc[3] = {coef0, coef1, coef2};
    int16_t* dc[3] = {nullptr, nullptr, nullptr};
    uint16_t* dq = nullptr;
    uint8_t* dout = nullptr;
    const size_t out_bytes = static_cast<size_t>(height) * stride;
    hipError_t e = hipSuccess;
    for (int c = 0; c < n_components && e == hipSuccess; ++c) {
        const size_t bytes = static_cast<size_t>(g.bw[c]) * g.bh[c] * 128u;
        e = hipMalloc(reinterpret_cast<void**>(&dc[c]), bytes);
        if (e == hipSuccess) e = hipMemcpy(dc[c], hc[c], bytes, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dq), 128u * n_components);
    if (e == hipSuccess) e = hipMemcpy(dq, qt, 128u * n_components, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), out_bytes);
    if (e == hipSuccess) e = hipMemcpy(dout, bgra, static_cast<size_t>(height - 1) * stride + static_cast<size_t>(width) * 4u, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = ifhip_jpeg_idct_color_batch_device(stage, dc[0], dc[1], dc[2], dq, 1, dout, out_bytes, stride, nullptr);
        if (rc == IFHIP_OK) {
            e = hipStreamSynchronize(nullptr);
            if (e == hipSuccess)
                e = hipMemcpy(bgra, dout, static_cast<size_t>(height - 1) * stride + static_cast<size_t>(width) * 4u, hipMemcpyDeviceToHost);
        }
    }
    for (auto* p : dc) if (p) (void)hipFree(p);
    if (dq) (void)hipFree(dq);
    if (dout) (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: jpeg stage staging failed: %s", hipGetErrorString(e));
    return IFHIP_OK;
}

}  // extern "C"

// ==================================================================================================================
// 8x8 -> NxN spatial block scalers (c_components/lib/codecs_jpeg_idct_fast.c flow_scale_spatial[_srgb]_NxN)
// ==================================================================================================================
#include "block_scalers.hpp"

namespace ifhip {

struct ScalerArgs {
    const uint8_t* in;       // plane of 8x8 blocks
    uint8_t* out;            // plane of NxN blocks
    uint32_t in_pitch, out_pitch, blocks_w, blocks_h;
    uint32_t n;
    int srgb;
    int32_t w[7][8];
    uint32_t log2_div[7];
    const uint16_t* s2l;     // 256 x u16 (12-bit linear)
    const uint8_t* l2s;      // 4096 x u8
};

// one lane per (block, output row r): vertical pass for the 8 columns, then the N horizontal outputs -- the order
// the generated C uses ("Scale vertically, then horizontally", variation.rs:240); all int32, exact.
__global__ void __launch_bounds__(256) scale_spatial_kernel(const ScalerArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = a.blocks_w * a.blocks_h * a.n;
    if (t >= total) return;
    const uint32_t r = t % a.n, b = t / a.n;
    const uint32_t by = b / a.blocks_w, bx = b - by * a.blocks_w;
    const uint8_t* blk = a.in + static_cast<size_t>(by) * 8u * a.in_pitch + bx * 8u;
    int32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t wr = a.w[r][i];
        const uint2 row = *reinterpret_cast<const uint2*>(blk + static_cast<size_t>(i) * a.in_pitch);
        const uint32_t px[2] = {row.x, row.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t byte = (px[j >> 2] >> (8 * (j & 3))) & 255u;
            const int32_t p = a.srgb ? static_cast<int32_t>(a.s2l[byte]) : static_cast<int32_t>(byte);
            v[j] += wr * p;
        }
    }
    uint8_t* orow = a.out + static_cast<size_t>(by * a.n + r) * a.out_pitch + bx * a.n;
    for (uint32_t c = 0; c < a.n; ++c) {
        const uint32_t sh = a.log2_div[r] + a.log2_div[c];
        int32_t sum = static_cast<int32_t>(1u << (sh - 1u));             // divisor_sum / 2
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j] * a.w[c][j];
        uint32_t o;
        if (sum < 0) o = 0;
        else if (static_cast<uint32_t>(sum) >= (4096u << sh)) o = 255;     // REVERSE_LUT_SIZE_SHORT * divisor_sum
        else o = a.srgb ? a.l2s[sum >> sh] : static_cast<uint32_t>(sum >> sh);
        orow[c] = static_cast<uint8_t>(o);
    }
}

}  // namespace ifhip


extern "C" {

int ifhip_block_scaler_tables(int n, int8_t* weights_7x8, uint8_t* log2_divisors_7, uint16_t* srgb_to_linear_256,
                              uint8_t* linear_to_srgb_4096) {
    const BlockScalerTables* t = block_scaler_tables();
    if (!t) return fail(IFHIP_INVALID_STATE, "InvalidState: block scaler tables could not be generated");
    if (n < 1 || n > 7) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: block scaler size %d", n);
    if (weights_7x8) std::memcpy(weights_7x8, t->scaler[n].w, 56);
    if (log2_divisors_7) std::memcpy(log2_divisors_7, t->scaler[n].log2_div, 7);
    if (srgb_to_linear_256) std::memcpy(srgb_to_linear_256, t->srgb_to_linear, 512);
    if (linear_to_srgb_4096) std::memcpy(linear_to_srgb_4096, t->linear_to_srgb, 4096);
    return IFHIP_OK;
}

int ifhip_scale_spatial_plane_device(const uint8_t* d_in, uint32_t in_pitch, uint32_t blocks_w, uint32_t blocks_h,
                                     int n, int srgb, uint8_t* d_out, uint32_t out_pitch, void* hip_stream) {
    if (n < 1 || n > 7) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: block scaler size %d", n);
    if (!d_in || !d_out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null plane pointer");
    if (blocks_w == 0 || blocks_h == 0) return IFHIP_OK;
    if ((in_pitch & 7u) || (reinterpret_cast<uintptr_t>(d_in) & 7u) || in_pitch < blocks_w * 8u || out_pitch < blocks_w * static_cast<uint32_t>(n))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: block planes need 8-byte aligned rows and pitch >= row bytes");
    ScalerDeviceTables dt;
    int rc = scaler_device_tables(&dt);
    if (rc) return rc;
    const BlockScalerTables* t = block_scaler_tables();
    ScalerArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = d_in; a.out = d_out; a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.blocks_w = blocks_w; a.blocks_h = blocks_h;
    a.n = static_cast<uint32_t>(n); a.srgb = srgb ? 1 : 0; a.s2l = dt.s2l; a.l2s = dt.l2s;
    for (int i = 0; i < 7; ++i) {
        a.log2_div[i] = t->scaler[n].log2_div[i];
        for (int j = 0; j < 8; ++j) a.w[i][j] = t->scaler[n].w[i][j];
    }
    const uint64_t total = static_cast<uint64_t>(blocks_w) * blocks_h * static_cast<uint32_t>(n);
    if (total > 0x7fffffffull) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: plane too large for one launch");
    hipLaunchKernelGGL(scale_spatial_kernel, dim3(static_cast<uint32_t>((total + 255u) / 256u)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

int ifhip_scale_spatial_blocks(const uint8_t* blocks, uint32_t n_blocks, int n, int srgb, uint8_t* out) {
    if (n < 1 || n > 7) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: block scaler size %d", n);
    if (n_blocks == 0) return IFHIP_OK;
    if (!blocks || !out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null block pointer");
    uint8_t *d_in = nullptr, *d_out = nullptr;
    const size_t in_bytes = static_cast<size_t>(n_blocks) * 64u, out_bytes = static_cast<size_t>(n_blocks) * n * n;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), in_bytes));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), out_bytes);
    int rc = IFHIP_OK;
    if (e == hipSuccess) e = hipMemcpy(d_in, blocks, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // the block array is a plane one block wide: pitch 8 in, pitch n out
        rc = ifhip_scale_spatial_plane_device(d_in, 8, 1, n_blocks, n, srgb, d_out, static_cast<uint32_t>(n), nullptr);
        if (rc == IFHIP_OK) {
            e = hipStreamSynchronize(nullptr);
            if (e == hipSuccess) e = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: block scaler staging failed: %s", hipGetErrorString(e));
    return IFHIP_OK;
}

}  // extern "C"
