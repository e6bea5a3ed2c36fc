// resample_kernels.hip -- the generic two-pass resample kernels, the flatten kernel and the launch dispatch.
// (The fused kernel lives in resample_fused.hip; shared device code in resample_device.hpp.)
#include <atomic>

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

constexpr uint32_t kHpassRows = 16;      // output rows per workgroup of the generic horizontal pass (amortises its table fill)
template <bool ALPHA>
__global__ void __launch_bounds__(256)
hpass_generic_kernel(const ResampleArgs a, const float4* scratch, uint32_t img0) {
    // The output stage's encode table in LDS: its three dependent lookups per pixel in global memory were most of this
    // kernel's time.  One fill (16 KiB) serves 256 columns x kHpassRows rows.
    __shared__ __attribute__((aligned(16))) uint8_t l2s_lds[16384];
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) reinterpret_cast<uint4*>(l2s_lds)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    __syncthreads();
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t img = img0 + blockIdx.z;
    if (u >= a.out_w) return;
    const uint32_t left = a.h_left[u], n = a.h_count[u];
    const float* w = a.h_w + a.h_off[u];
    const OutTables<const float*, const uint8_t*> tb{a.lut_in, l2s_lds};
    const uint32_t j_end = min((blockIdx.y + 1u) * kHpassRows, a.out_h);
    for (uint32_t j = blockIdx.y * kHpassRows; j < j_end; ++j) {
        const float4* row = scratch + (static_cast<size_t>(blockIdx.z) * a.out_h + j) * a.in_w;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (uint32_t k = 0; k < n; ++k) {
            const float4 v = row[left + k];
            const float wk = w[k];
            s0 = __builtin_fmaf(wk, v.x, s0);
            s1 = __builtin_fmaf(wk, v.y, s1);
            s2 = __builtin_fmaf(wk, v.z, s2);
            if (ALPHA) s3 = __builtin_fmaf(wk, v.w, s3);
        }
        store_pixel<ALPHA>(a, img, j, u, s0, s1, s2, ALPHA ? s3 : 1.0f, tb);
    }
}

// ------------------------------------------------------------------------------------------------------
// Banded two-pass kernel: the generic pair fused through LDS, for shapes whose source window per band of output rows is
// small -- up-scales (each output row reads 2 .. 6 source rows; the fused kernel's ring would need one slot per output
// row a source row feeds: 3x Robidoux has 15, its limit is 8) and small frames.  One workgroup = (band of R output rows,
// every G-th frame): tables once per workgroup, then per frame the band's source rows are converted ONCE into LDS
// (working floats, premultiplied), every output row of the band is filtered vertically from them into LDS, then
// horizontally into the output stage.  The f32 intermediate the generic pair writes to and reads from HBM
// (out_h x in_w x 16 bytes per frame) never exists, every source byte is converted once per band instead of once per
// tap, and the next frame's source pixels are requested while the current frame is filtered.  Same arithmetic as the
// generic kernels term for term (vpass_generic_kernel / hpass_generic_kernel above: the same ascending fmaf chains on
// the same values), hence bit-identical to them and to the oracle.  Vertical pass: rows over waves (their weights are
// wave-uniform: scalar loads), columns over lanes.  Horizontal pass: a lane keeps its output column over the wave's rows.
// Wide frames (a band of whole rows does not fit the LDS: 960 -> 1920 columns) are cut into COLUMN STRIPS as well: a workgroup
// = (band, strip of strip_w output columns, every G-th frame), its LDS rows hold the strip's source columns only.
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t kBandedThreads = 512;
constexpr uint32_t kBandedTableBytes = 16384 + 1024 + 16;   // linear -> sRGB table, sRGB -> float table, a strip's ranges
constexpr uint32_t kBandedPrefetch = 8;                     // source pixels a lane keeps in flight for the next frame
template <bool ALPHA>
__global__ void __launch_bounds__(kBandedThreads)
banded_resample_kernel(const ResampleArgs a, const BandedArgs b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    uint8_t* l2s_lds = bsm;
    float* s2f = reinterpret_cast<float*>(bsm + 16384);
    const bool h_lds = (b.flags & 4u) != 0u;
    uint32_t* hl = reinterpret_cast<uint32_t*>(bsm + kBandedTableBytes);               // [strip_w] left, count, weight offset; weights
    uint32_t* hc = hl + b.strip_w;
    uint32_t* ho = hc + b.strip_w;
    float* hw = reinterpret_cast<float*>(ho + b.strip_w);
    const uint32_t h_bytes = h_lds ? ((3u * b.strip_w + b.h_w_floats) * 4u + 15u) & ~15u : 0u;
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    const uint32_t lane = tid & 63u, nw = T >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);                   // wave-uniform: rows and their tables go the scalar way
    const uint32_t tiles = b.n_bands * b.n_strips;
    const uint32_t tile = blockIdx.x % tiles, g = blockIdx.x / tiles;
    const uint32_t band = tile % b.n_bands, strip = tile / b.n_bands;
    const uint32_t j0 = band * b.rows_per_band, j1 = min(j0 + b.rows_per_band, a.out_h), nrows = j1 - j0;
    // the strip's output columns [u0, u1), its source columns [cx0, cx0 + sw) and its slice [wo0, wo0 + nwf) of the weights:
    // the union of its columns' windows (trimmed zero weights move a window's ends: no column is taken for the first or last)
    const uint32_t u0 = strip * b.strip_w, u1 = min(u0 + b.strip_w, a.out_w), nu = u1 - u0;
    uint32_t cx0 = 0u, sw = a.in_w, wo0 = 0u, nwf = b.h_w_floats;
    if (b.n_strips > 1u) {
        uint32_t* rng = reinterpret_cast<uint32_t*>(bsm + 16384 + 1024);            // {min left, max right, min offset, max offset end}
        if (tid < 4u) rng[tid] = (tid & 1u) ? 0u : 0xffffffffu;
        __syncthreads();
        uint32_t lo = 0xffffffffu, hi = 0u, wlo = 0xffffffffu, whi = 0u;
        for (uint32_t i = tid; i < nu; i += T) {
            const uint32_t l = a.h_left[u0 + i], c = a.h_count[u0 + i], o = a.h_off[u0 + i];
            lo = min(lo, l); hi = max(hi, l + c); wlo = min(wlo, o); whi = max(whi, o + c);
        }
        if (tid < nu) { atomicMin(&rng[0], lo); atomicMax(&rng[1], hi); atomicMin(&rng[2], wlo); atomicMax(&rng[3], whi); }
        __syncthreads();
        cx0 = rng[0]; sw = rng[1] - cx0; wo0 = rng[2]; nwf = rng[3] - wo0;
    }
    float4* src = reinterpret_cast<float4*>(bsm + kBandedTableBytes + h_bytes);        // [src rows][sw]
    float4* vband = src + static_cast<size_t>(b.src_rows_cap) * sw;                    // [rows_per_band][sw]
    for (uint32_t i = tid; i < 1024u; i += T) reinterpret_cast<uint4*>(l2s_lds)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    for (uint32_t i = tid; i < 256u; i += T) s2f[i] = a.lut_in[i];
    if (h_lds) {
        for (uint32_t i = tid; i < nu; i += T) { hl[i] = a.h_left[u0 + i] - cx0; hc[i] = a.h_count[u0 + i]; ho[i] = a.h_off[u0 + i] - wo0; }
        for (uint32_t i = tid; i < nwf; i += T) hw[i] = a.h_w[wo0 + i];
    }
    uint32_t row0, row1;                                     // the band's source rows [row0, row1): at most src_rows_cap (host)
    if (b.flags & 2u) {
        row0 = a.v_left[j0];
        row1 = a.v_left[j1 - 1u] + a.v_count[j1 - 1u];
    } else {
        row0 = 0xffffffffu; row1 = 0u;
        for (uint32_t j = j0; j < j1; ++j) {
            const uint32_t l = a.v_left[j];
            row0 = min(row0, l);
            row1 = max(row1, l + a.v_count[j]);
        }
    }
    const uint32_t npx = (row1 - row0) * sw;                 // source pixels of the tile, LDS index r * sw + x
    // Source pixels this lane converts (LDS index tid + c * T): their byte offsets inside a frame, the same for every frame.
    const bool prefetch = npx <= kBandedPrefetch * T;
    uint32_t off[kBandedPrefetch], pre[kBandedPrefetch];
#pragma unroll
    for (uint32_t c = 0; c < kBandedPrefetch; ++c) {
        const uint32_t i = tid + c * T;
        const uint32_t r = i / sw, x = i - r * sw;
        off[c] = (row0 + r) * a.in_stride + 4u * (cx0 + x);
        pre[c] = 0u;
    }
    auto request = [&](uint32_t img) {
        const uint8_t* frame = a.in + static_cast<size_t>(img) * a.in_image_bytes;
#pragma unroll
        for (uint32_t c = 0; c < kBandedPrefetch; ++c)
            if (tid + c * T < npx) pre[c] = *reinterpret_cast<const uint32_t*>(frame + off[c]);
    };
    auto to_float = [&](uint32_t px) -> float4 {             // arithmetic contract step 1, as vpass_generic_kernel
        float f0 = s2f[px & 255u], f1 = s2f[(px >> 8) & 255u], f2 = s2f[(px >> 16) & 255u], f3 = 1.0f;
        if (ALPHA) {
            f3 = static_cast<float>(px >> 24) * (1.0f / 255.0f);
            f0 = f0 * f3; f1 = f1 * f3; f2 = f2 * f3;
        }
        return make_float4(f0, f1, f2, f3);
    };
    uint32_t img = g;
    if (prefetch && img < a.n_images) request(img);
    __syncthreads();                                         // tables
    const OutTables<const float*, const uint8_t*> tb{s2f, l2s_lds};
    for (; img < a.n_images; img += b.frame_step) {
        // ---- sample -> working float, once per source pixel of the band ----
        if (prefetch) {
#pragma unroll
            for (uint32_t c = 0; c < kBandedPrefetch; ++c)
                if (tid + c * T < npx) src[tid + c * T] = to_float(pre[c]);
            if (img + b.frame_step < a.n_images) request(img + b.frame_step);      // in flight under both passes below
        } else {
            const uint8_t* frame = a.in + static_cast<size_t>(img) * a.in_image_bytes;
            for (uint32_t r = wave; r < row1 - row0; r += nw) {
                const uint32_t* prow = reinterpret_cast<const uint32_t*>(frame + static_cast<size_t>(row0 + r) * a.in_stride);
                for (uint32_t x = lane; x < sw; x += 64u) src[r * sw + x] = to_float(prow[cx0 + x]);
            }
        }
        __syncthreads();     // src complete; every wave is also done with the previous frame's horizontal pass (vband is free)
        // ---- vertical pass of every output row of the band (step 2) ----
        for (uint32_t jr = wave; jr < nrows; jr += nw) {
            const uint32_t j = j0 + jr;
            const uint32_t left = a.v_left[j] - row0, n = a.v_count[j];
            const float* w = a.v_w + a.v_off[j];             // (wave-uniform address: scalar loads)
            for (uint32_t x = lane; x < sw; x += 64u) {
                const float4* col = src + left * sw + x;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                for (uint32_t k = 0; k < n; ++k) {
                    const float4 v = col[k * sw];
                    const float wk = w[k];
                    s0 = __builtin_fmaf(wk, v.x, s0);
                    s1 = __builtin_fmaf(wk, v.y, s1);
                    s2 = __builtin_fmaf(wk, v.z, s2);
                    if (ALPHA) s3 = __builtin_fmaf(wk, v.w, s3);
                }
                vband[jr * sw + x] = make_float4(s0, s1, s2, ALPHA ? s3 : 1.0f);
            }
        }
        __syncthreads();     // vband complete; src is free for the next frame
        // ---- horizontal pass + output stage (step 3 and the compositing modes).  A lane keeps its output column over the
        // wave's rows.  (Weights of short windows kept in registers, padded to 8 taps, measured slower -- 3.76 against 3.65 ms
        // on the 3x shape -- and were removed.)
        for (uint32_t ui = lane; ui < nu; ui += 64u) {
            const uint32_t u = u0 + ui;
            const uint32_t left = h_lds ? hl[ui] : a.h_left[u] - cx0, n = h_lds ? hc[ui] : a.h_count[u];
            const uint32_t woff = h_lds ? ho[ui] : a.h_off[u];
            auto weight = [&](uint32_t k) -> float { return h_lds ? hw[woff + k] : a.h_w[woff + k]; };
            for (uint32_t jr = wave; jr < nrows; jr += nw) {
                const float4* row = vband + jr * sw;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                for (uint32_t k = 0; k < n; ++k) {
                    const float4 v = row[left + k];
                    const float wk = weight(k);
                    s0 = __builtin_fmaf(wk, v.x, s0);
                    s1 = __builtin_fmaf(wk, v.y, s1);
                    s2 = __builtin_fmaf(wk, v.z, s2);
                    if (ALPHA) s3 = __builtin_fmaf(wk, v.w, s3);
                }
                store_pixel<ALPHA>(a, img, j0 + jr, u, s0, s1, s2, ALPHA ? s3 : 1.0f, tb);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Flatten: graphics/blend.rs:6-59, in place.  One lane = 4 adjacent pixels x kMatteRows rows (16-byte accesses when the
// rows are 16-byte aligned).  Both tables live in LDS: the sRGB->linear table with one copy per bank (conflict-free for
// any pixel values, as in the resample kernel), the 16 KiB linear->sRGB table as is; 512 lanes x 32 rows per workgroup
// amortise the 48 KiB table fill (from L2) over 256 KiB of pixels.
// ------------------------------------------------------------------------------------------------------
struct MatteArgs {
    uint8_t* bgra;
    size_t image_bytes;
    uint32_t w, h, stride, n_images;
    uint32_t matte;             // B,G,R,A bytes
    float mb, mg, mr, ma;       // linear matte colour, matte alpha / 255
    const float* s2l;
    const uint8_t* l2s;
    uint32_t vec16;             // rows are 16-byte aligned
};
constexpr uint32_t kMatteRows = 32;
constexpr uint32_t kMatteLanes = 512;

__device__ __forceinline__ uint32_t matte_pixel(const MatteArgs& a, const BankedLut& lut, const uint8_t* l2s, uint32_t px) {
    const uint32_t pa = px >> 24;
    if (pa == 255u) return px;
    if (pa == 0u) return a.matte;
    const float paf = static_cast<float>(static_cast<int>(pa)) * (1.0f / 255.0f);
    const float ma = (1.0f - paf) * a.ma;
    const float fa = ma + paf;
    auto enc = [&](float v) -> uint32_t {
        const float s = __builtin_fminf(__builtin_fmaxf(v * 16383.0f, 0.0f), 16383.0f);     // NaN -> 0, as lut.rs:4-8
        return l2s[static_cast<uint32_t>(s)];
    };
    const uint32_t nb = enc((lut[px & 255u] * paf + a.mb * ma) / fa);
    const uint32_t ng = enc((lut[(px >> 8) & 255u] * paf + a.mg * ma) / fa);
    const uint32_t nr = enc((lut[(px >> 16) & 255u] * paf + a.mr * ma) / fa);
    const uint32_t na = uchar_clamp_ff(255.0f * fa);
    return nb | (ng << 8) | (nr << 16) | (na << 24);
}

__global__ void __launch_bounds__(kMatteLanes) apply_matte_kernel(const MatteArgs a) {
    __shared__ float lut_banked[256 * 32];
    __shared__ __attribute__((aligned(16))) uint8_t l2s[16384];
    for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) lut_banked[i] = a.s2l[i >> 5];
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x)
        reinterpret_cast<uint4*>(l2s)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    __syncthreads();
    const BankedLut lut{lut_banked, threadIdx.x & 31u, 5u};
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (x0 >= a.w) return;
    const uint32_t img = blockIdx.z;
    const uint32_t y_end = min((blockIdx.y + 1u) * kMatteRows, a.h);
    const bool full = a.vec16 && x0 + 3u < a.w;
    for (uint32_t yy = blockIdx.y * kMatteRows; yy < y_end; ++yy) {
        uint32_t* p = reinterpret_cast<uint32_t*>(a.bgra + static_cast<size_t>(img) * a.image_bytes
                                                  + static_cast<size_t>(yy) * a.stride) + x0;
        if (full) {
            uint4 v = *reinterpret_cast<uint4*>(p);
            if ((v.x & v.y & v.z & v.w) >> 24 == 255u) continue;            // four opaque pixels: nothing to do
            v.x = matte_pixel(a, lut, l2s, v.x); v.y = matte_pixel(a, lut, l2s, v.y);
            v.z = matte_pixel(a, lut, l2s, v.z); v.w = matte_pixel(a, lut, l2s, v.w);
            *reinterpret_cast<uint4*>(p) = v;
        } else {
            const uint32_t n = min(4u, a.w - x0);
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t px = p[i];
                if ((px >> 24) != 255u) p[i] = matte_pixel(a, lut, l2s, px);
            }
        }
    }
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
    const dim3 bh(256), gh((a.out_w + 255u) / 256u, (a.out_h + kHpassRows - 1u) / kHpassRows, n_img);
    if (alpha) {
        hipLaunchKernelGGL((vpass_generic_kernel<true>), gv, bv, 0, st, a, scratch, img0);
        hipLaunchKernelGGL((hpass_generic_kernel<true>), gh, bh, 0, st, a, scratch, img0);
    } else {
        hipLaunchKernelGGL((vpass_generic_kernel<false>), gv, bv, 0, st, a, scratch, img0);
        hipLaunchKernelGGL((hpass_generic_kernel<false>), gh, bh, 0, st, a, scratch, img0);
    }
    return hipGetLastError();
}

hipError_t launch_banded(const ResampleArgs& a, bool alpha, const BandedArgs& b, uint32_t grid_x, size_t lds, hipStream_t st) {
    // the dynamic-LDS cap is sticky per kernel and device: raised once for each (a bit per device ordinal; ordinals beyond
    // the mask's width simply set the attribute again on every launch), and a failure to raise it is the launch's error
    static std::atomic<uint64_t> raised[2];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<uint64_t>& r = raised[alpha ? 1 : 0];
    const uint64_t bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (!(r.load(std::memory_order_relaxed) & bit)) {
        const void* f = alpha ? reinterpret_cast<const void*>(&banded_resample_kernel<true>) : reinterpret_cast<const void*>(&banded_resample_kernel<false>);
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsCap));
        if (e != hipSuccess) return e;
        r.fetch_or(bit, std::memory_order_relaxed);
    }
    const dim3 grid(grid_x), block(kBandedThreads);
    if (alpha) hipLaunchKernelGGL((banded_resample_kernel<true>), grid, block, lds, st, a, b);
    else hipLaunchKernelGGL((banded_resample_kernel<false>), grid, block, lds, st, a, b);
    return hipGetLastError();
}

hipError_t launch_apply_matte(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                              uint32_t stride, uint32_t matte, float mb, float mg, float mr, float ma,
                              const float* s2l, const uint8_t* l2s, hipStream_t st) {
    // grid.y / grid.z are 16-bit: tall bitmaps and big batches go out as several launches
    const uint32_t rows_per_launch = 65535u * kMatteRows;
    for (uint32_t i0 = 0; i0 < n_images; i0 += 65535u) {
        const uint32_t n = min(65535u, n_images - i0);
        for (uint32_t y0 = 0; y0 < h; y0 += rows_per_launch) {
            const uint32_t hh = min(rows_per_launch, h - y0);
            uint8_t* base = d_bgra + static_cast<size_t>(i0) * image_bytes + static_cast<size_t>(y0) * stride;
            MatteArgs m{base, image_bytes, w, hh, stride, n, matte, mb, mg, mr, ma, s2l, l2s, 0u};
            m.vec16 = ((reinterpret_cast<uintptr_t>(base) | image_bytes | stride) & 15u) == 0 ? 1u : 0u;
            const dim3 block(kMatteLanes), grid(((w + 3u) / 4u + kMatteLanes - 1u) / kMatteLanes, (hh + kMatteRows - 1u) / kMatteRows, n);
            hipLaunchKernelGGL(apply_matte_kernel, grid, block, 0, st, m);
        }
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Read-only streaming probe: the yardstick for a kernel whose traffic is (almost) all reads.  Same access shape as
// the resample kernel's source rows -- 16-byte non-temporal loads, consecutive lanes on consecutive 16-byte groups,
// each workgroup walking its own contiguous span -- with one XOR per load as the only arithmetic.
// ------------------------------------------------------------------------------------------------------
typedef uint32_t probe_vec_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void read_probe_kernel(const probe_vec_t* __restrict__ src, size_t n_vec, uint32_t* sink) {
    const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = per_wg * blockIdx.x;
    size_t hi = lo + per_wg;
    if (hi > n_vec) hi = n_vec;
    uint32_t acc = 0;
    size_t i = lo + threadIdx.x;
    for (; i + 3u * 1024u < hi; i += 4u * 1024u) {
        const probe_vec_t a = __builtin_nontemporal_load(src + i);
        const probe_vec_t b = __builtin_nontemporal_load(src + i + 1024u);
        const probe_vec_t c = __builtin_nontemporal_load(src + i + 2048u);
        const probe_vec_t d = __builtin_nontemporal_load(src + i + 3072u);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < hi; i += 1024u) {
        const probe_vec_t a = __builtin_nontemporal_load(src + i);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) sink[threadIdx.x] = acc;          // practically never true: keeps the loads alive
}

// The same with writes mixed in: one 16-byte vector per lane stored for every `every` vectors it read (the resample
// kernels of the moderate ratios write 15 - 36 % of their bytes).  The store goes through inline asm for the reason the
// canvas stores do (resample_device.hpp): loads and stores share vmcnt, and a store the compiler tracks drains the loads
// in flight.  sink[1024] receives the stores one lane of workgroup 0 made (every workgroup makes as many, give or take one).
__global__ __launch_bounds__(1024) void mix_probe_kernel(const probe_vec_t* __restrict__ src, probe_vec_t* __restrict__ dst, size_t n_vec,
                                                         uint32_t every, uint32_t* sink) {
    const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = per_wg * blockIdx.x;
    size_t hi = lo + per_wg;
    if (hi > n_vec) hi = n_vec;
    uint32_t acc = 0, since = 0, stores = 0;
    size_t w = lo + threadIdx.x;
    typedef __attribute__((address_space(1))) probe_vec_t gout;
    for (size_t i = lo + threadIdx.x; i + 3u * 1024u < hi; i += 4u * 1024u) {
        const probe_vec_t a = __builtin_nontemporal_load(src + i);
        const probe_vec_t b = __builtin_nontemporal_load(src + i + 1024u);
        const probe_vec_t c = __builtin_nontemporal_load(src + i + 2048u);
        const probe_vec_t d = __builtin_nontemporal_load(src + i + 3072u);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
        since += 4u;
        while (since >= every) {                                 // (wave-uniform)
            since -= every;
            const probe_vec_t ov = {acc, acc, acc, acc};
            gout* op = reinterpret_cast<gout*>(reinterpret_cast<uintptr_t>(dst + w));
            asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(op), "v"(ov) : "memory");
            w += 1024u;
            ++stores;
        }
    }
    if (acc == 0x9e3779b9u) sink[threadIdx.x] = acc;
    if (blockIdx.x == 0u && threadIdx.x == 0u) sink[1024] = stores;
}

hipError_t launch_mix_probe(const uint8_t* d, uint8_t* out, size_t bytes, uint32_t every, uint32_t* sink, hipStream_t st) {
    hipLaunchKernelGGL(mix_probe_kernel, dim3(256u * 8u), dim3(1024), 0, st, reinterpret_cast<const probe_vec_t*>(d),
                       reinterpret_cast<probe_vec_t*>(out), bytes / 16u, every, sink);
    return hipGetLastError();
}

hipError_t launch_read_probe(const uint8_t* d, size_t bytes, uint32_t* sink, hipStream_t st) {
    hipLaunchKernelGGL(read_probe_kernel, dim3(256u * 8u), dim3(1024), 0, st, reinterpret_cast<const probe_vec_t*>(d), bytes / 16u, sink);
    return hipGetLastError();
}

}  // namespace ifhip
