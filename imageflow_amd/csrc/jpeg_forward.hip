// jpeg_forward.hip -- gfx950 kernels + C ABI for the encode-side JPEG pixel stage.
//
// Replaces the arithmetic libjpeg runs between write_scanlines and the entropy coder when MozjpegEncoder::write_frame
// (codecs/mozjpeg.rs:78-160, classic preset: set_fastest_defaults -> no trellis, no deringing; JDCT_ISLOW; input
// JCS_EXT_BGRA / JCS_EXT_BGRX) compresses a frame: rgb_ycc_convert (jccolor.c), h2v1 / h2v2 down-sampling with edge
// expansion (jcsample.c, jcprepct.c), jpeg_fdct_islow (jfdctint.c), quantisation (jcdctmgr.c) and the dummy blocks of
// the last MCU column / row (jccoefct.c).  All integer; coefficient-exact against oracle/jpeg_oracle.c jo_jpeg_forward,
// which is pinned to the coefficients libjpeg-turbo wrote into tests/golden/jpeg_encode_cases.npz.
//
// One fused kernel (bound: HBM; byte work, no MFMA): 4 B/px read once, 2 B per coefficient written once
// (4:2:0: 7 B/px, 4:4:4: 10 B/px); nothing else touches HBM.
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>

#include "common.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace ifhip {

struct FwdGeom {
    uint32_t width, height;
    uint32_t hs[3], vs[3], hmax, vmax;
    uint32_t bw[3], bh[3];          // coefficient blocks per row / column, MCU padded
    uint32_t rbw[3], rbh[3];        // real blocks (ceil(downsampled size / 8)); the rest are dummy blocks
    uint32_t dw[3], dh[3];          // downsampled component size
};

struct FwdArgs {
    FwdGeom g;
    const uint8_t* bgra;
    size_t image_bytes;
    uint32_t stride, n_images;
    uint32_t vec16;                 // every pixel row starts 16-byte aligned
    const uint16_t* qt;             // [n_images][3][64]
    int16_t* coef[3];               // [n_images][bh][bw][64]
};

constexpr int kFwdBlocksPerWg = 32;
constexpr int kFwdBlockPitch = 72;  // dwords per 8x8 workspace in LDS (64 + 8: spreads 4 blocks over the 32 banks)

__device__ __forceinline__ int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// jccolor.c rgb_ycc_convert, 16-bit fixed point, evaluated in place; px = B | G<<8 | R<<16 | A<<24
__device__ __forceinline__ int32_t rgb_to_component(uint32_t px, uint32_t c) {
    const int32_t b = px & 255u, g = (px >> 8) & 255u, r = (px >> 16) & 255u;
    if (c == 0u) return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    if (c == 1u) return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

// jfdctint.c jpeg_fdct_islow, one 8-point pass; first = row pass (outputs scaled up by PASS1_BITS).
// The reference factorisation (12 multiplies shared through z1..z5) is exact integer arithmetic, so every output is an
// integer linear combination of the butterflies; distributing the 13-bit constants gives per-output coefficient pairs
// that fit int16, and v_dot2_i32_i16 evaluates them (with the descale rounding constant as accumulator) bit-identically:
//   o2 = 10703*t13 + 4433*t12            o6 = 4433*t13 - 10704*t12
//   o7 = -11363*t4 + 9633*t5 - 6436*t6 + 2260*t7      o5 = 9633*t4 + 2261*t5 - 11362*t6 + 6437*t7
//   o3 = -6436*t4 - 11362*t5 - 2259*t6 + 9633*t7      o1 = 2260*t4 + 6437*t5 + 9633*t6 + 11363*t7
// Operand ranges: row pass |t| <= 510; column pass |t12|,|t13| <= 16.4k, |t4..7| <= 8.2k (row outputs are <= 4096).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pack16(int32_t lo, int32_t hi) { return s16x2{static_cast<short>(lo), static_cast<short>(hi)}; }
__device__ __forceinline__ int32_t dot2(s16x2 a, short c0, short c1, int32_t acc) {
    return __builtin_amdgcn_sdot2(a, s16x2{c0, c1}, acc, false);
}

__device__ __forceinline__ void fdct8(const int32_t (&e)[8], int32_t (&o)[8], bool first) {
    const int32_t tmp0 = e[0] + e[7], tmp7 = e[0] - e[7], tmp1 = e[1] + e[6], tmp6 = e[1] - e[6];
    const int32_t tmp2 = e[2] + e[5], tmp5 = e[2] - e[5], tmp3 = e[3] + e[4], tmp4 = e[3] - e[4];
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int sh = first ? 11 : 15;                      // CONST_BITS -/+ PASS1_BITS
    const int32_t rnd = 1 << (sh - 1);
    if (first) {
        o[0] = static_cast<int32_t>(static_cast<uint32_t>(tmp10 + tmp11) << 2);
        o[4] = static_cast<int32_t>(static_cast<uint32_t>(tmp10 - tmp11) << 2);
    } else {
        o[0] = (tmp10 + tmp11 + 2) >> 2;
        o[4] = (tmp10 - tmp11 + 2) >> 2;
    }
    const s16x2 ev = pack16(tmp13, tmp12), od0 = pack16(tmp4, tmp5), od1 = pack16(tmp6, tmp7);
    o[2] = dot2(ev, 10703, 4433, rnd) >> sh;
    o[6] = dot2(ev, 4433, -10704, rnd) >> sh;
    o[7] = dot2(od1, -6436, 2260, dot2(od0, -11363, 9633, rnd)) >> sh;
    o[5] = dot2(od1, -11362, 6437, dot2(od0, 9633, 2261, rnd)) >> sh;
    o[3] = dot2(od1, -2259, 9633, dot2(od0, -6436, -11362, rnd)) >> sh;
    o[1] = dot2(od1, 9633, 11363, dot2(od0, 2260, 6437, rnd)) >> sh;
}

// jcdctmgr.c quantize: round-half-up division of |v| by 8*Q, sign restored.  The quotient is estimated with one f32
// multiply and corrected by the exact integer remainder (|v| + 4Q < 2^24, so the estimate is off by at most one).
// Everything that depends on Q alone comes from the workgroup's table: e = (8Q - 1, bits of 1 / 8Q, 4Q, -8Q).  No compare
// + select pairs: on gfx950 a VALU compare into an SGPR pair costs two wait states before the select may read it, and this
// runs 64 times per block.
__device__ __forceinline__ int32_t quantize_coef(int32_t v, const uint4& e) {
    const int32_t s = v >> 31;                                   // 0 / -1
    const int32_t u = ((v ^ s) - s) + static_cast<int32_t>(e.z); // |v| + 4Q
    int32_t q = static_cast<int32_t>(static_cast<uint32_t>(static_cast<float>(static_cast<uint32_t>(u)) * __uint_as_float(e.y)));
    const int32_t r = __mul24(q, static_cast<int32_t>(e.w)) + u; // u - q * 8Q (both factors < 2^24)
    q += (r >> 31) - ((static_cast<int32_t>(e.x) - r) >> 31);    // r < 0: one too many; r >= 8Q: one too few
    return (q ^ s) - s;
}

__device__ __forceinline__ void wave_lds_sync() {       // the 8 lanes of a block share a wave: order LDS traffic, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup = one tile of 256 x (8*VS) pixels of one frame.
// Phase 1: every BGRA pixel is read once (16-byte loads), converted, down-sampled and parked in LDS as bytes (Y rows
//   pitch 288, chroma rows pitch CW+32: both conflict-free for the 8-byte row reads of phase 2).
// Phase 2: 32 blocks per pass, 8 lanes per block: row pass (lane = row), column pass + quantisation (lane = column),
//   16-byte coefficient rows out (a wave writes 8 consecutive blocks = 1 KiB).
// Dummy blocks are written as zeros; their DC is patched by jpeg_dummy_dc_kernel.
template <int HS, int VS>
__global__ void __launch_bounds__(256) jpeg_forward_fused_kernel(const FwdArgs a) {
    constexpr int TW = 256, YP = 288, CW = TW / HS, CP = CW + 32;
    constexpr int NPASS = VS + (HS == 2 ? 1 : 2);
    __shared__ __attribute__((aligned(16))) uint8_t ys[8 * VS * YP];
    __shared__ __attribute__((aligned(16))) uint8_t cs[2][8 * CP];
    __shared__ int32_t ws[kFwdBlocksPerWg * kFwdBlockPitch];
    // what the quantiser needs of the image's three tables (quantize_coef): the reciprocal is a quarter-rate instruction and
    // depends on the table entry only -- once per workgroup instead of once per coefficient
    __shared__ uint4 qtab[3 * 64];
    const uint32_t t = threadIdx.x, img = blockIdx.z;
    const uint32_t W = a.g.width, H = a.g.height;
    const uint8_t* src = a.bgra + static_cast<size_t>(img) * a.image_bytes;
    if (t < 192u) {
        const int32_t qv = static_cast<int32_t>(a.qt[static_cast<size_t>(img) * 192u + t]) << 3;
        qtab[t] = make_uint4(static_cast<uint32_t>(qv - 1), __float_as_uint(__builtin_amdgcn_rcpf(static_cast<float>(qv))),   // 1 ulp is enough
                             static_cast<uint32_t>(qv >> 1), static_cast<uint32_t>(-qv));
    }

    // ---- phase 1: colour conversion + down-sampling.  Edge expansion = clamped source coordinates (jcsample.c
    //      expand_right_edge, jcprepct.c expand_bottom_edge); chroma rows past the last down-sampled row repeat it ----
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t item = t + 256u * k, cg = item & 63u, cy = item >> 6;
        const uint32_t x0 = blockIdx.x * TW + cg * 4u, gcy = blockIdx.y * 8u + cy;
        uint32_t rows[VS];
        bool past = false;
        if (VS == 2) {
            const uint32_t dhc = a.g.dh[1];
            past = gcy >= dhc;
            const uint32_t c2 = past ? dhc - 1u : gcy;
            rows[0] = min(2u * c2, H - 1u);
            rows[VS - 1] = min(2u * c2 + 1u, H - 1u);
        } else {
            rows[0] = min(gcy, H - 1u);
        }
        uint32_t px[VS][4];
#pragma unroll
        for (int r = 0; r < VS; ++r) {
            const uint8_t* row = src + static_cast<size_t>(rows[r]) * a.stride;
            if (a.vec16 && x0 + 3u < W) {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(row + static_cast<size_t>(x0) * 4u));
                px[r][0] = v.x; px[r][1] = v.y; px[r][2] = v.z; px[r][3] = v.w;
            } else {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i)
                    px[r][i] = *reinterpret_cast<const uint32_t*>(row + static_cast<size_t>(min(x0 + i, W - 1u)) * 4u);
            }
        }
#pragma unroll
        for (int r = 0; r < VS; ++r) {       // luma rows 2*gcy + r, clamped: past the end both are the last picture row
            const int sr = (VS == 2 && r == 0 && past) ? VS - 1 : r;
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) packed |= static_cast<uint32_t>(rgb_to_component(px[sr][i], 0u)) << (8u * i);
            *reinterpret_cast<uint32_t*>(&ys[(cy * VS + r) * YP + cg * 4u]) = packed;
        }
#pragma unroll
        for (uint32_t c = 1; c <= 2u; ++c) {
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u / HS; ++j) {
                int32_t sum = 0;
#pragma unroll
                for (int r = 0; r < VS; ++r)
#pragma unroll
                    for (uint32_t i = 0; i < static_cast<uint32_t>(HS); ++i) sum += rgb_to_component(px[r][j * HS + i], c);
                int32_t v;
                if (HS == 1) v = sum;
                else if (VS == 1) v = (sum + static_cast<int32_t>(j & 1u)) >> 1;        // h2v1: bias 0,1,0,1 (x0/2 is even)
                else v = (sum + 1 + static_cast<int32_t>(j & 1u)) >> 2;                 // h2v2: bias 1,2,1,2
                packed |= static_cast<uint32_t>(v & 255) << (8u * j);
            }
            uint8_t* dst = &cs[c - 1u][cy * CP + cg * (4u / HS)];
            if (HS == 1) *reinterpret_cast<uint32_t*>(dst) = packed;
            else *reinterpret_cast<uint16_t*>(dst) = static_cast<uint16_t>(packed);
        }
    }
    __syncthreads();

    // ---- phase 2: forward DCT + quantisation ----
    const uint32_t lane8 = t & 7u, lb = t >> 3;
    int32_t* w = ws + lb * kFwdBlockPitch;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        uint32_t c, bx, by, pitch;
        const uint8_t* sp;
        if (p < VS) {
            c = 0u; bx = blockIdx.x * 32u + lb; by = blockIdx.y * VS + p;
            sp = ys + (p * 8) * YP + lb * 8u; pitch = YP;
        } else if (HS == 2) {
            c = 1u + (lb >> 4); bx = blockIdx.x * 16u + (lb & 15u); by = blockIdx.y;
            sp = cs[lb >> 4] + (lb & 15u) * 8u; pitch = CP;
        } else {
            c = 1u + (p - VS); bx = blockIdx.x * 32u + lb; by = blockIdx.y;
            sp = cs[p - VS] + lb * 8u; pitch = CP;
        }
        const bool on = bx < a.g.bw[c];
        const bool real = on && bx < a.g.rbw[c] && by < a.g.rbh[c];
        uint4 outv = make_uint4(0, 0, 0, 0);
        if (real) {
            const uint2 v = *reinterpret_cast<const uint2*>(sp + lane8 * pitch);
            int32_t e[8], o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[k] = static_cast<int32_t>((v.x >> (8 * k)) & 255u) - 128;
                e[4 + k] = static_cast<int32_t>((v.y >> (8 * k)) & 255u) - 128;
            }
            fdct8(e, o, true);
#pragma unroll
            for (int k = 0; k < 8; ++k) w[lane8 * 8u + k] = o[k];
        }
        wave_lds_sync();
        int32_t o[8];
        if (real) {
            int32_t e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) e[r] = w[r * 8 + lane8];
            fdct8(e, o, false);
        }
        wave_lds_sync();
        if (real) {
            const uint4* q = qtab + c * 64u + lane8;
#pragma unroll
            for (int r = 0; r < 8; ++r) w[r * 8 + lane8] = quantize_coef(o[r], q[r * 8]);
        }
        wave_lds_sync();
        if (real) {
            const int32_t* r = w + lane8 * 8u;
            outv.x = (static_cast<uint32_t>(r[0]) & 0xffffu) | (static_cast<uint32_t>(r[1]) << 16);
            outv.y = (static_cast<uint32_t>(r[2]) & 0xffffu) | (static_cast<uint32_t>(r[3]) << 16);
            outv.z = (static_cast<uint32_t>(r[4]) & 0xffffu) | (static_cast<uint32_t>(r[5]) << 16);
            outv.w = (static_cast<uint32_t>(r[6]) & 0xffffu) | (static_cast<uint32_t>(r[7]) << 16);
        }
        wave_lds_sync();
        if (on) {
            int16_t* dst = a.coef[c] + ((static_cast<size_t>(img) * a.g.bh[c] + by) * a.g.bw[c] + bx) * 64u + lane8 * 8u;
            *reinterpret_cast<uint4*>(dst) = outv;
        }
    }
}

// jccoefct.c compress_data: dummy blocks at the right / bottom MCU edges carry AC = 0 and the DC of the previous block.
// Only luma can have them (chroma is sampled 1x1, so its MCU holds one block).
__global__ void __launch_bounds__(256) jpeg_dummy_dc_kernel(const FwdArgs a) {
    const uint32_t c = 0u, img = blockIdx.y;
    const uint32_t bidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (bidx >= a.g.bw[c] * a.g.bh[c]) return;
    const uint32_t by = bidx / a.g.bw[c], bx = bidx - by * a.g.bw[c];
    if (bx < a.g.rbw[c] && by < a.g.rbh[c]) return;
    int16_t* base = a.coef[c] + static_cast<size_t>(img) * a.g.bw[c] * a.g.bh[c] * 64u;
    uint32_t sy, sx;
    if (by < a.g.rbh[c]) { sy = by; sx = a.g.rbw[c] - 1u; }                       // right edge: the row's last real block
    else {                                                                        // bottom edge: previous block of the MCU
        sy = a.g.rbh[c] - 1u;
        const uint32_t prev = (bx / a.g.hs[c]) * a.g.hs[c] + a.g.hs[c] - 1u;
        sx = prev < a.g.rbw[c] ? prev : a.g.rbw[c] - 1u;
    }
    base[(static_cast<size_t>(by) * a.g.bw[c] + bx) * 64u] = base[(static_cast<size_t>(sy) * a.g.bw[c] + sx) * 64u];
}

}  // namespace ifhip

using namespace ifhip;

struct ifhip_jpeg_fwd_stage {
    int device = -1;
    FwdGeom g;
    uint32_t max_images = 0;
};

static int make_fwd_geom(uint32_t width, uint32_t height, const uint8_t* hs, const uint8_t* vs, FwdGeom* g) {
    if (width == 0 || height == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Bitmap dimensions cannot be zero");
    if (!hs || !vs) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null sampling factors");
    std::memset(g, 0, sizeof *g);
    g->width = width; g->height = height;
    for (int c = 0; c < 3; ++c) { g->hs[c] = hs[c]; g->vs[c] = vs[c]; }
    g->hmax = hs[0]; g->vmax = vs[0];
    const bool ok = hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1 && (g->hmax == 1 || g->hmax == 2) &&
                    (g->vmax == 1 || g->vmax == 2) && !(g->hmax == 1 && g->vmax == 2);
    if (!ok) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: only 4:4:4, 4:2:2 (h2v1) and 4:2:0 sampling");
    const uint32_t mw = (width + 8u * g->hmax - 1u) / (8u * g->hmax), mh = (height + 8u * g->vmax - 1u) / (8u * g->vmax);
    for (int c = 0; c < 3; ++c) {
        g->bw[c] = mw * g->hs[c]; g->bh[c] = mh * g->vs[c];
        g->dw[c] = (width * g->hs[c] + g->hmax - 1u) / g->hmax;
        g->dh[c] = (height * g->vs[c] + g->vmax - 1u) / g->vmax;
        g->rbw[c] = (g->dw[c] + 7u) / 8u; g->rbh[c] = (g->dh[c] + 7u) / 8u;
    }
    return IFHIP_OK;
}

extern "C" {

int ifhip_jpeg_fwd_stage_create(ifhip_jpeg_fwd_stage** stage, uint32_t width, uint32_t height, const uint8_t* h_samp,
                                const uint8_t* v_samp, uint32_t max_images) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage out-pointer");
    *stage = nullptr;
    if (max_images == 0 || max_images > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: 1..65535 images per stage");
    std::unique_ptr<ifhip_jpeg_fwd_stage> s(new ifhip_jpeg_fwd_stage);
    int rc = make_fwd_geom(width, height, h_samp, v_samp, &s->g);
    if (rc) return rc;
    if (s->g.bh[0] > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: more than 65535 block rows per launch");
    if (int arc = require_gfx950(&s->device)) return arc;
    s->max_images = max_images;
    *stage = s.release();
    return IFHIP_OK;
}

void ifhip_jpeg_fwd_stage_destroy(ifhip_jpeg_fwd_stage* stage) { delete stage; }

int ifhip_jpeg_fwd_stage_block_dims(const ifhip_jpeg_fwd_stage* stage, uint32_t* blocks_w3, uint32_t* blocks_h3) {
    if (!stage || !blocks_w3 || !blocks_h3) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    for (int c = 0; c < 3; ++c) { blocks_w3[c] = stage->g.bw[c]; blocks_h3[c] = stage->g.bh[c]; }
    return IFHIP_OK;
}

int ifhip_jpeg_forward_batch_device(ifhip_jpeg_fwd_stage* stage, const uint8_t* d_bgra, size_t image_bytes, uint32_t stride,
                                    const uint16_t* d_qt, uint32_t n_images, int16_t* d_coef0, int16_t* d_coef1,
                                    int16_t* d_coef2, void* hip_stream) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage");
    if (n_images == 0) return IFHIP_OK;
    if (n_images > stage->max_images) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: %u images exceed the stage capacity %u", n_images, stage->max_images);
    if (!d_bgra || !d_qt || !d_coef0 || !d_coef1 || !d_coef2) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    if (static_cast<uint64_t>(stage->g.width) * 4u > stride || (stride & 3u) || (image_bytes & 3u) || (reinterpret_cast<uintptr_t>(d_bgra) & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: BGRA rows must be 4-byte aligned and stride >= 4*w");
    if ((reinterpret_cast<uintptr_t>(d_coef0) | reinterpret_cast<uintptr_t>(d_coef1) | reinterpret_cast<uintptr_t>(d_coef2)) & 15u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient planes must be 16-byte aligned");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != stage->device) return fail(IFHIP_INVALID_STATE, "InvalidState: stage belongs to device %d, current device is %d", stage->device, dev);
    FwdArgs a;
    std::memset(&a, 0, sizeof a);
    a.g = stage->g; a.bgra = d_bgra; a.image_bytes = image_bytes; a.stride = stride; a.n_images = n_images; a.qt = d_qt;
    a.coef[0] = d_coef0; a.coef[1] = d_coef1; a.coef[2] = d_coef2;
    a.vec16 = ((reinterpret_cast<uintptr_t>(d_bgra) | image_bytes | stride) & 15u) == 0 ? 1u : 0u;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t mh = a.g.bh[0] / a.g.vs[0];
    const dim3 grid((a.g.bw[0] * 8u + 255u) / 256u, mh, n_images);
    if (a.g.hmax == 1) hipLaunchKernelGGL((jpeg_forward_fused_kernel<1, 1>), grid, dim3(256), 0, st, a);
    else if (a.g.vmax == 1) hipLaunchKernelGGL((jpeg_forward_fused_kernel<2, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((jpeg_forward_fused_kernel<2, 2>), grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    if (a.g.rbw[0] != a.g.bw[0] || a.g.rbh[0] != a.g.bh[0]) {
        hipLaunchKernelGGL(jpeg_dummy_dc_kernel, dim3((a.g.bw[0] * a.g.bh[0] + 255u) / 256u, n_images), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    return IFHIP_OK;
}

int ifhip_jpeg_forward(const uint8_t* bgra, uint32_t width, uint32_t height, uint32_t stride, const uint8_t* h_samp,
                       const uint8_t* v_samp, const uint16_t* qt, int16_t* coef0, int16_t* coef1, int16_t* coef2) {
    if (!bgra || !qt || !coef0 || !coef1 || !coef2) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    ifhip_jpeg_fwd_stage* stage = nullptr;
    int rc = ifhip_jpeg_fwd_stage_create(&stage, width, height, h_samp, v_samp, 1);
    if (rc) return rc;
    std::unique_ptr<ifhip_jpeg_fwd_stage> guard(stage);
    if (static_cast<uint64_t>(width) * 4u > stride || (stride & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: stride smaller than a BGRA row or not a multiple of 4");
    const size_t in_bytes = static_cast<size_t>(height) * stride, in_valid = static_cast<size_t>(height - 1) * stride + static_cast<size_t>(width) * 4u;
    uint8_t* d_in = nullptr;
    uint16_t* d_qt = nullptr;
    int16_t* d_c[3] = {nullptr, nullptr, nullptr};
    int16_t* h_c[3] = {coef0, coef1, coef2};
    size_t cb[3] = {0, 0, 0};
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), in_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, bgra, in_valid, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_qt), 384);
    if (e == hipSuccess) e = hipMemcpy(d_qt, qt, 384, hipMemcpyHostToDevice);
    for (int c = 0; c < 3 && e == hipSuccess; ++c) {
        cb[c] = static_cast<size_t>(stage->g.bw[c]) * stage->g.bh[c] * 128u;
        e = hipMalloc(reinterpret_cast<void**>(&d_c[c]), cb[c]);
    }
    if (e == hipSuccess) {
        rc = ifhip_jpeg_forward_batch_device(stage, d_in, in_bytes, stride, d_qt, 1, d_c[0], d_c[1], d_c[2], nullptr);
        if (rc == IFHIP_OK) {
            e = hipStreamSynchronize(nullptr);
            for (int c = 0; c < 3 && e == hipSuccess; ++c) e = hipMemcpy(h_c[c], d_c[c], cb[c], hipMemcpyDeviceToHost);
        }
    }
    if (d_in) (void)hipFree(d_in);
    if (d_qt) (void)hipFree(d_qt);
    for (auto* p : d_c) if (p) (void)hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: jpeg forward staging failed: %s", hipGetErrorString(e));
    return IFHIP_OK;
}

}  // extern "C"
