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

#include "device.hpp"

namespace ifhip {

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

__device__ __forceinline__ uint8_t encode_channel(const ResampleArgs& a, float v) {   // color.rs:61-71
    if (a.linear) {                                                                   // lut.rs:4-8
        float s = v * 16383.0f;
        s = (s != s) ? 0.0f : s;
        s = s < 0.0f ? 0.0f : s;
        s = s > 16383.0f ? 16383.0f : s;
        return a.l2s[static_cast<uint32_t>(s)];
    }
    return uchar_clamp_ff(255.0f * v);
}

// px: premultiplied working-space pixel (B,G,R,A).  Returns the BGRA8 word to store at the canvas pixel
// whose current content is `dst` (only read for BlendWithSelf).
template <bool ALPHA>
__device__ __forceinline__ uint32_t render_pixel(const ResampleArgs& a, float p0, float p1, float p2, float pa,
                                                 uint32_t dst, const float* lut) {
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

template <bool ALPHA>
__device__ __forceinline__ void store_pixel(const ResampleArgs& a, uint32_t img, uint32_t j, uint32_t u,
                                            float p0, float p1, float p2, float pa, const float* lut) {
    uint8_t* cp = a.canvas + static_cast<size_t>(img) * a.canvas_image_bytes
                  + static_cast<size_t>(a.y + j) * a.c_stride + static_cast<size_t>(a.x + u) * 4u;
    uint32_t* cw = reinterpret_cast<uint32_t*>(cp);           // canvas rows are 4-byte aligned (checked on host)
    uint32_t dst = 0;
    if (ALPHA && a.mode == IFHIP_BLEND_WITH_SELF) dst = *cw;
    *cw = render_pixel<ALPHA>(a, p0, p1, p2, pa, dst, lut);
    if (a.f32_dump) {
        float4* d = reinterpret_cast<float4*>(a.f32_dump) + (static_cast<size_t>(img) * a.out_h + j) * a.out_w + u;
        *d = make_float4(p0, p1, p2, ALPHA ? pa : 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------------
// Fused kernel: one workgroup = (image, band of output rows, column strip)
// ------------------------------------------------------------------------------------------------------
constexpr int kPrefetch = 2;        // source rows in flight per lane beyond the one being consumed

template <int K, bool ALPHA>
__global__ void __launch_bounds__(1024)
fused_resample_kernel(const ResampleArgs a) {
    constexpr int C = ALPHA ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t T = blockDim.x;
    uint32_t b = blockIdx.x;
    const uint32_t strip_i = b % a.n_strips; b /= a.n_strips;
    const uint32_t band = b % a.n_bands;
    const uint32_t img = b / a.n_bands;

    const Strip strip = a.strips[strip_i];
    const uint32_t n_u = strip.u1 - strip.u0;

    float* lut = reinterpret_cast<float*>(smem);                 // 256 floats
    float* obuf = lut + 256;                                     // n_u * 4 floats
    float* inter = obuf + ((n_u * 4u + 3u) & ~3u);               // nquads * 4 * C floats

    for (uint32_t i = tid; i < 256u; i += T) lut[i] = a.lut_in[i];
    __syncthreads();

    const uint32_t s0 = a.band_begin[band], s1 = a.band_begin[band + 1];
    const bool lane_on = tid < strip.nquads;
    const uint8_t* src = a.in + static_cast<size_t>(img) * a.in_image_bytes
                         + static_cast<size_t>(strip.cx0 + 4u * tid) * 4u;

    float acc[K][4][C];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[s][p][c] = 0.0f;

    auto fetch = [&](uint32_t si) -> uint4 {
        uint4 r = make_uint4(0, 0, 0, 0);
        if (si < s1) {
            const int y = a.steps[si].y;                          // wave-uniform
            if (y >= 0 && lane_on)
                r = *reinterpret_cast<const uint4*>(src + static_cast<size_t>(y) * a.in_stride);
        }
        return r;
    };

    uint4 raw[kPrefetch];
#pragma unroll
    for (int d = 0; d < kPrefetch; ++d) raw[d] = fetch(s0 + d);

    for (uint32_t sb = s0; sb < s1; sb += kPrefetch) {
#pragma unroll
        for (int d = 0; d < kPrefetch; ++d) {
            const uint32_t si = sb + d;
            if (si >= s1) break;
            const VStep st = a.steps[si];                        // 64-byte scalar load
            const uint4 cur = raw[d];
            raw[d] = fetch(si + kPrefetch);

            if (st.y >= 0) {
                const uint32_t w4[4] = {cur.x, cur.y, cur.z, cur.w};
                float v[4][C];
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
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st.active & (1u << s)) {                 // scalar branch
                        const float w = st.w[s];
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[s][p][c] = __builtin_fmaf(w, v[p][c], acc[s][p][c]);
                    }
                }
            }

            if (st.flush_slot >= 0) {
                // ---- hand the finished vertically-filtered row to the horizontal pass through LDS ----
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st.flush_slot == s) {
                        if (lane_on) {
                            float* dstp = inter + static_cast<size_t>(tid) * (4 * C);
#pragma unroll
                            for (int p = 0; p < 4; ++p)
#pragma unroll
                                for (int c = 0; c < C; ++c) dstp[p * C + c] = acc[s][p][c];
                        }
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[s][p][c] = 0.0f;
                    }
                }
                __syncthreads();
                const uint32_t j = static_cast<uint32_t>(st.out_row);
                const uint32_t n_chain = n_u * C;
                for (uint32_t idx = tid; idx < n_chain; idx += T) {
                    const uint32_t ul = idx / C, c = idx - ul * C, u = strip.u0 + ul;
                    const uint32_t left = a.h_left[u] - strip.cx0, n = a.h_count[u];
                    const float* wp = a.h_wT + u;
                    const float* ip = inter + static_cast<size_t>(left) * C + c;
                    float h = 0.0f;
                    for (uint32_t k = 0; k < n; ++k) h = __builtin_fmaf(wp[static_cast<size_t>(k) * a.out_w], ip[k * C], h);
                    obuf[ul * 4u + c] = h;
                }
                __syncthreads();
                for (uint32_t ul = tid; ul < n_u; ul += T) {
                    const float* o = obuf + ul * 4u;
                    store_pixel<ALPHA>(a, img, j, strip.u0 + ul, o[0], o[1], o[2], ALPHA ? o[3] : 1.0f, lut);
                }
            }
        }
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
    store_pixel<ALPHA>(a, img, j, u, s0, s1, s2, ALPHA ? s3 : 1.0f, a.lut_in);
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
template <int K>
static hipError_t launch_fused_k(const ResampleArgs& a, bool alpha, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (alpha) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_resample_kernel<K, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((fused_resample_kernel<K, true>), grid, block, lds, st, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_resample_kernel<K, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((fused_resample_kernel<K, false>), grid, block, lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_fused(const ResampleArgs& a, int slots, bool alpha, uint32_t grid, uint32_t block, size_t lds,
                        hipStream_t st) {
    const dim3 g(grid), b(block);
    switch (slots) {
    case 1: case 2: return launch_fused_k<2>(a, alpha, g, b, lds, st);
    case 3: case 4: return launch_fused_k<4>(a, alpha, g, b, lds, st);
    case 5: return launch_fused_k<5>(a, alpha, g, b, lds, st);
    case 6: return launch_fused_k<6>(a, alpha, g, b, lds, st);
    case 7: case 8: return launch_fused_k<8>(a, alpha, g, b, lds, st);
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
