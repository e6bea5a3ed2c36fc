// resample_kernels.hip -- the generic two-pass resample kernels, the flatten kernel and the launch dispatch.
// (The fused kernel lives in resample_fused.hip; shared device code in resample_device.hpp.)
#include "resample_device.hpp"

namespace ifhip {

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
#define IFHIP_DECL_K(n) hipError_t launch_fused_k##n(const ResampleArgs&, bool, bool, dim3, dim3, size_t, hipStream_t);
IFHIP_DECL_K(1) IFHIP_DECL_K(2) IFHIP_DECL_K(3) IFHIP_DECL_K(4) IFHIP_DECL_K(5) IFHIP_DECL_K(6) IFHIP_DECL_K(7) IFHIP_DECL_K(8)

hipError_t launch_fused(const ResampleArgs& a, int slots, bool alpha, bool per_pixel, uint32_t grid, uint32_t block, size_t lds,
                        hipStream_t st) {
    const dim3 g(grid), b(block);
    switch (slots) {            // one translation unit per ring size (resample_fused.hip)
    case 1: return launch_fused_k1(a, alpha, per_pixel, g, b, lds, st);
    case 2: return launch_fused_k2(a, alpha, per_pixel, g, b, lds, st);
    case 3: return launch_fused_k3(a, alpha, per_pixel, g, b, lds, st);
    case 4: return launch_fused_k4(a, alpha, per_pixel, g, b, lds, st);
    case 5: return launch_fused_k5(a, alpha, per_pixel, g, b, lds, st);
    case 6: return launch_fused_k6(a, alpha, per_pixel, g, b, lds, st);
    case 7: return launch_fused_k7(a, alpha, per_pixel, g, b, lds, st);
    case 8: return launch_fused_k8(a, alpha, per_pixel, g, b, lds, st);
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
