// bitmap_ops.hip -- gfx950 kernels + C ABI for the whole-bitmap byte operations that surround the resampler in
// imageflow's graphs, so that decode -> orient -> crop -> resize -> watermark ->
// encode chains never leave HBM:
//   ifhip_apply_color_matrix*   graphics/color_matrix.rs:5-29   (ColorFilterSrgb / watermark opacity, flow/nodes/color.rs)
//   ifhip_copy_rect*            graphics/copy_rect.rs:12-119    (crop, clone, expand_canvas, copy_rect_to_canvas)
//   ifhip_fill_rect*            graphics/bitmaps.rs:1504-1548   (fill_rect, expand_canvas background)
//   ifhip_flip_vertical* / ifhip_flip_horizontal*   graphics/flip.rs:10-38
//   ifhip_transpose*            graphics/transpose.rs:95-121    (transpose, rotate 90/270, EXIF orientations 5..8)
// All HBM-bound byte work (no MFMA): 16-byte accesses where the rectangle allows, an LDS tile for the transpose.
// Bit-exact against oracle/bitmap_oracle.c.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

#include "common.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace ifhip {

struct Frames {                 // a batch of equally shaped BGRA8 frames
    uint8_t* base;
    size_t image_bytes;
    uint32_t w, h, stride;
};

__device__ __forceinline__ uint32_t* px_ptr(const Frames& f, uint32_t img, uint32_t x, uint32_t y) {
    return reinterpret_cast<uint32_t*>(f.base + static_cast<size_t>(img) * f.image_bytes + static_cast<size_t>(y) * f.stride) + x;
}

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

struct Matrix5 { float m[25]; };

// color_matrix.rs:5-29: one lane = 4 adjacent pixels x kCmRows rows (16-byte accesses when the rows allow); the sums are
// evaluated left to right with one rounding per operation (this translation unit is compiled with -ffp-contract=off,
// like the oracle).
constexpr uint32_t kCmRows = 8;

__device__ __forceinline__ uint32_t color_matrix_pixel(const float* m, float m40, float m41, float m42, float m43, uint32_t px) {
    const float b = static_cast<float>(px & 255u), g = static_cast<float>((px >> 8) & 255u);
    const float r = static_cast<float>((px >> 16) & 255u), a = static_cast<float>(px >> 24);
    const uint32_t nr = uchar_clamp_ff(m[0] * r + m[5] * g + m[10] * b + m[15] * a + m40);
    const uint32_t ng = uchar_clamp_ff(m[1] * r + m[6] * g + m[11] * b + m[16] * a + m41);
    const uint32_t nb = uchar_clamp_ff(m[2] * r + m[7] * g + m[12] * b + m[17] * a + m42);
    const uint32_t na = uchar_clamp_ff(m[3] * r + m[8] * g + m[13] * b + m[18] * a + m43);
    return nb | (ng << 8) | (nr << 16) | (na << 24);
}

__global__ void __launch_bounds__(256) color_matrix_kernel(const Frames f, const Matrix5 k, const uint32_t vec16) {
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (x0 >= f.w) return;
    const float* m = k.m;
    const float m40 = m[20] * 255.0f, m41 = m[21] * 255.0f, m42 = m[22] * 255.0f, m43 = m[23] * 255.0f;
    const uint32_t y_end = min((blockIdx.y + 1u) * kCmRows, f.h);
    const bool full = vec16 && x0 + 3u < f.w;
    for (uint32_t y = blockIdx.y * kCmRows; y < y_end; ++y) {
        uint32_t* p = px_ptr(f, blockIdx.z, x0, y);
        if (full) {
            uint4 v = *reinterpret_cast<uint4*>(p);
            v.x = color_matrix_pixel(m, m40, m41, m42, m43, v.x); v.y = color_matrix_pixel(m, m40, m41, m42, m43, v.y);
            v.z = color_matrix_pixel(m, m40, m41, m42, m43, v.z); v.w = color_matrix_pixel(m, m40, m41, m42, m43, v.w);
            *reinterpret_cast<uint4*>(p) = v;
        } else {
            const uint32_t n = min(4u, f.w - x0);
            for (uint32_t i = 0; i < n; ++i) p[i] = color_matrix_pixel(m, m40, m41, m42, m43, p[i]);
        }
    }
}

// Rectangle kernels: one lane per 4 adjacent pixels of the rectangle (16-byte accesses when `vec` says every group
// is aligned on both sides), 4-byte accesses for the ragged tail or unaligned rectangles.
struct RectArgs {
    Frames src, dst;
    uint32_t sx, sy, dx, dy, w, h;
    uint32_t vec;               // 16-byte accesses are legal for full groups
    uint32_t value;             // fill colour / OR mask
};

enum RectOp { kCopy = 0, kFill = 1, kOrMask = 2, kSwapRows = 3 };

template <int OP>
__global__ void __launch_bounds__(256) rect_kernel(const RectArgs a) {
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (x0 >= a.w) return;
    const uint32_t y = blockIdx.y, img = blockIdx.z;
    uint32_t* d = px_ptr(a.dst, img, a.dx + x0, a.dy + y);
    uint32_t* s = (OP == kCopy || OP == kSwapRows) ? px_ptr(a.src, img, a.sx + x0, a.sy + (OP == kSwapRows ? a.h * 2u - 1u - y + a.value : y)) : nullptr;
    const bool full = x0 + 3u < a.w;
    if (a.vec && full) {
        uint4* dv = reinterpret_cast<uint4*>(d);
        if (OP == kCopy) *dv = *reinterpret_cast<const uint4*>(s);
        else if (OP == kFill) *dv = make_uint4(a.value, a.value, a.value, a.value);
        else if (OP == kOrMask) { uint4 v = *dv; v.x |= a.value; v.y |= a.value; v.z |= a.value; v.w |= a.value; *dv = v; }
        else { uint4* sv = reinterpret_cast<uint4*>(s); const uint4 t = *dv; *dv = *sv; *sv = t; }
    } else {
        const uint32_t n = full ? 4u : a.w - x0;
        for (uint32_t i = 0; i < n; ++i) {
            if (OP == kCopy) d[i] = s[i];
            else if (OP == kFill) d[i] = a.value;
            else if (OP == kOrMask) d[i] |= a.value;
            else { const uint32_t t = d[i]; d[i] = s[i]; s[i] = t; }
        }
    }
}

// flip.rs:26-38: lane x < w/2 swaps pixels x and w-1-x of its row (both sides are contiguous runs per wave)
__global__ void __launch_bounds__(256) flip_h_kernel(const Frames f) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= f.w / 2u) return;
    uint32_t* row = px_ptr(f, blockIdx.z, 0, blockIdx.y);
    const uint32_t t = row[x];
    row[x] = row[f.w - 1u - x];
    row[f.w - 1u - x] = t;
}

// transpose.rs:95-121: 64x64 pixel tiles through LDS (pitch 65 dwords: conflict-free in both directions); a wave reads
// 256 contiguous bytes of a source row and writes 256 contiguous bytes of a destination row.
__global__ void __launch_bounds__(256) transpose_kernel(const Frames from, const Frames to) {
    __shared__ uint32_t tile[64][65];
    const uint32_t lx = threadIdx.x & 63u, ly = threadIdx.x >> 6, img = blockIdx.z;
    const uint32_t x0 = blockIdx.x * 64u, y0 = blockIdx.y * 64u;
#pragma unroll 4
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t y = y0 + i * 4u + ly, x = x0 + lx;
        if (x < from.w && y < from.h) tile[i * 4u + ly][lx] = *px_ptr(from, img, x, y);
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t ty = x0 + i * 4u + ly, tx = y0 + lx;          // destination row = source column
        if (tx < to.w && ty < to.h) *px_ptr(to, img, tx, ty) = tile[lx][i * 4u + ly];
    }
}

static int require_device() {
    static std::mutex mu;
    static int ok_device = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device; this library has no CPU path");
    std::lock_guard<std::mutex> lock(mu);
    if (dev == ok_device) return IFHIP_OK;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
    ok_device = dev;
    return IFHIP_OK;
}

static int check_frames(const void* p, size_t image_bytes, uint32_t w, uint32_t h, uint32_t stride, uint32_t n, const char* what) {
    if (!p) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null %s pointer", what);
    if (static_cast<uint64_t>(w) * 4u > stride || (stride & 3u) || (image_bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: %s rows must be 4-byte aligned and stride >= 4*w", what);
    if (h > 65535u || n > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: more than 65535 rows/images per launch");
    return IFHIP_OK;
}

static bool aligned16(const Frames& f, uint32_t x) {
    return ((reinterpret_cast<uintptr_t>(f.base) | f.image_bytes | f.stride | (static_cast<uintptr_t>(x) * 4u)) & 15u) == 0;
}

template <int OP>
static int launch_rect(const RectArgs& a, uint32_t rows, uint32_t n, hipStream_t st) {
    if (a.w == 0 || rows == 0 || n == 0) return IFHIP_OK;
    const dim3 grid(((a.w + 3u) / 4u + 255u) / 256u, rows, n);
    hipLaunchKernelGGL((rect_kernel<OP>), grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

static int set_alpha_255(const Frames& f, uint32_t n, hipStream_t st) {        // bitmaps.rs normalize_unused_alpha
    RectArgs a{};
    a.dst = f; a.w = f.w; a.h = f.h; a.value = 0xFF000000u; a.vec = aligned16(f, 0) ? 1u : 0u;
    return launch_rect<kOrMask>(a, f.h, n, st);
}

// host-buffer drop-ins: stage one bitmap through HBM around a device call
struct Staged {
    uint8_t* d = nullptr;
    size_t bytes = 0, valid = 0;
    ~Staged() { if (d) (void)hipFree(d); }
    int up(const uint8_t* host, uint32_t w, uint32_t h, uint32_t stride, bool copy) {
        if (w == 0 || h == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Bitmap dimensions cannot be zero");
        if (!host) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null bitmap pointer");
        if (static_cast<uint64_t>(w) * 4u > stride || (stride & 3u))
            return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: stride smaller than a BGRA row or not a multiple of 4");
        valid = static_cast<size_t>(h - 1) * stride + static_cast<size_t>(w) * 4u;
        bytes = (static_cast<size_t>(h) * stride + 15u) & ~static_cast<size_t>(15);
        int rc = require_device();
        if (rc) return rc;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), bytes));
        if (copy) HIP_TRY(hipMemcpy(d, host, valid, hipMemcpyHostToDevice));
        return IFHIP_OK;
    }
    int down(uint8_t* host) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(host, d, valid, hipMemcpyDeviceToHost));
        return IFHIP_OK;
    }
};

}  // namespace ifhip

using namespace ifhip;

extern "C" {

int ifhip_apply_color_matrix_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                          uint32_t stride, const float* matrix25, void* hip_stream) {
    if (!matrix25) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null matrix");
    if (w == 0 || h == 0 || n_images == 0) return IFHIP_OK;
    int rc = check_frames(d_bgra, image_bytes, w, h, stride, n_images, "bitmap");
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    Matrix5 k;
    std::memcpy(k.m, matrix25, sizeof k.m);
    const Frames f{d_bgra, image_bytes, w, h, stride};
    hipLaunchKernelGGL(color_matrix_kernel, dim3(((w + 3u) / 4u + 255u) / 256u, (h + kCmRows - 1u) / kCmRows, n_images), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), f, k, aligned16(f, 0) ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

int ifhip_apply_color_matrix(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, const float* matrix25) {
    Staged s;
    int rc = s.up(bgra, w, h, stride, true);
    if (rc) return rc;
    if ((rc = ifhip_apply_color_matrix_batch_device(s.d, s.bytes, 1, w, h, stride, matrix25, nullptr))) return rc;
    return s.down(bgra);
}

int ifhip_copy_rect_batch_device(uint8_t* d_in, size_t in_image_bytes, uint32_t in_w, uint32_t in_h, uint32_t in_stride,
                                 int in_alpha_meaningful, uint8_t* d_canvas, size_t canvas_image_bytes, uint32_t canvas_w,
                                 uint32_t canvas_h, uint32_t canvas_stride, int* canvas_alpha_meaningful, uint32_t from_x,
                                 uint32_t from_y, uint32_t to_x, uint32_t to_y, uint32_t w, uint32_t h, uint32_t n_images,
                                 void* hip_stream) {
    if (!canvas_alpha_meaningful) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null canvas_alpha_meaningful");
    if (in_w <= from_x || in_h <= from_y || static_cast<uint64_t>(in_w) < static_cast<uint64_t>(from_x) + w ||
        static_cast<uint64_t>(in_h) < static_cast<uint64_t>(from_y) + h || static_cast<uint64_t>(canvas_w) < static_cast<uint64_t>(to_x) + w ||
        static_cast<uint64_t>(canvas_h) < static_cast<uint64_t>(to_y) + h)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Invalid argument to copy_rect. Canvas is %ux%u, Input is %ux%u, Arguments provided: (%u, %u, %u, %u, %u, %u)",
                    canvas_w, canvas_h, in_w, in_h, from_x, from_y, to_x, to_y, w, h);
    int rc = check_frames(d_in, in_image_bytes, in_w, in_h, in_stride, n_images, "input");
    if (rc) return rc;
    if ((rc = check_frames(d_canvas, canvas_image_bytes, canvas_w, canvas_h, canvas_stride, n_images, "canvas"))) return rc;
    if (d_in == d_canvas) return fail(IFHIP_INVALID_ARGUMENT, "InvalidNodeConnections: Canvas and Input are the same bitmap!");
    if ((rc = require_device())) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const Frames in{d_in, in_image_bytes, in_w, in_h, in_stride}, cv{d_canvas, canvas_image_bytes, canvas_w, canvas_h, canvas_stride};
    if (!*canvas_alpha_meaningful && in_alpha_meaningful) {          // copy_rect.rs:47-54
        if ((rc = set_alpha_255(cv, n_images, st))) return rc;
        *canvas_alpha_meaningful = 1;
    }
    if (!in_alpha_meaningful && *canvas_alpha_meaningful)            // copy_rect.rs:64-66 (the input bitmap is normalised too)
        if ((rc = set_alpha_255(in, n_images, st))) return rc;
    RectArgs a{};
    a.src = in; a.dst = cv; a.sx = from_x; a.sy = from_y; a.dx = to_x; a.dy = to_y; a.w = w; a.h = h;
    a.vec = (aligned16(in, from_x) && aligned16(cv, to_x)) ? 1u : 0u;
    return launch_rect<kCopy>(a, h, n_images, st);
}

int ifhip_fill_rect_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h, uint32_t stride,
                                 int compositing, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t color_bgra,
                                 void* hip_stream) {
    if (compositing == IFHIP_BLEND_WITH_MATTE && !(x1 == 0 && y1 == 0 && x2 == w && y2 == h))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Cannot draw a rectangle on a sub-rectangle of a bitmap in BlendWithMatte mode");
    if (y2 == y1 || x2 == x1) return IFHIP_OK;                       // "Don't fail on zero width rect"
    if (y2 <= y1 || x2 <= x1 || x2 > w || y2 > h)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Coordinates %u,%u %u,%u must be within image dimensions %ux%u", x1, y1, x2, y2, w, h);
    int rc = check_frames(d_bgra, image_bytes, w, h, stride, n_images, "bitmap");
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    RectArgs a{};
    a.dst = Frames{d_bgra, image_bytes, w, h, stride};
    a.dx = x1; a.dy = y1; a.w = x2 - x1; a.h = y2 - y1; a.value = color_bgra; a.vec = aligned16(a.dst, x1) ? 1u : 0u;
    return launch_rect<kFill>(a, a.h, n_images, static_cast<hipStream_t>(hip_stream));
}

int ifhip_normalize_unused_alpha_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                              uint32_t stride, int alpha_meaningful, void* hip_stream) {
    if (alpha_meaningful) return IFHIP_OK;                           // bitmaps.rs:1571-1573
    if (w == 0 || h == 0 || n_images == 0) return IFHIP_OK;
    int rc = check_frames(d_bgra, image_bytes, w, h, stride, n_images, "bitmap");
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    return set_alpha_255(Frames{d_bgra, image_bytes, w, h, stride}, n_images, static_cast<hipStream_t>(hip_stream));
}

int ifhip_flip_vertical_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                     uint32_t stride, void* hip_stream) {
    if (w == 0 || h == 0 || n_images == 0) return IFHIP_OK;
    int rc = check_frames(d_bgra, image_bytes, w, h, stride, n_images, "bitmap");
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    RectArgs a{};
    a.src = a.dst = Frames{d_bgra, image_bytes, w, h, stride};
    a.w = w; a.h = h / 2u;                 // row y of the top half <-> row h-1-y:  sy + (2*(h/2) - 1 - y + value) = h-1-y
    a.value = h - 2u * (h / 2u);           // 1 for odd heights (the middle row stays)
    a.vec = aligned16(a.dst, 0) ? 1u : 0u;
    return launch_rect<kSwapRows>(a, h / 2u, n_images, static_cast<hipStream_t>(hip_stream));
}

int ifhip_flip_horizontal_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                       uint32_t stride, void* hip_stream) {
    if (w < 2u || h == 0 || n_images == 0) return IFHIP_OK;
    int rc = check_frames(d_bgra, image_bytes, w, h, stride, n_images, "bitmap");
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    const Frames f{d_bgra, image_bytes, w, h, stride};
    hipLaunchKernelGGL(flip_h_kernel, dim3((w / 2u + 255u) / 256u, h, n_images), dim3(256), 0, static_cast<hipStream_t>(hip_stream), f);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

int ifhip_transpose_batch_device(const uint8_t* d_from, size_t from_image_bytes, uint32_t from_w, uint32_t from_h,
                                 uint32_t from_stride, uint8_t* d_to, size_t to_image_bytes, uint32_t to_w, uint32_t to_h,
                                 uint32_t to_stride, uint32_t n_images, void* hip_stream) {
    if (from_w != to_h || from_h != to_w)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: For transposition, canvas and input formats must be the same and dimensions must be swapped");
    if (from_w == 0 || from_h == 0 || n_images == 0) return IFHIP_OK;
    int rc = check_frames(d_from, from_image_bytes, from_w, from_h, from_stride, n_images, "input");
    if (rc) return rc;
    if ((rc = check_frames(d_to, to_image_bytes, to_w, to_h, to_stride, n_images, "canvas"))) return rc;
    if (d_from == d_to) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Canvas and input must be different bitmaps for transpose to work!");
    if ((rc = require_device())) return rc;
    const Frames f{const_cast<uint8_t*>(d_from), from_image_bytes, from_w, from_h, from_stride}, t{d_to, to_image_bytes, to_w, to_h, to_stride};
    const dim3 grid((from_w + 63u) / 64u, (from_h + 63u) / 64u, n_images);
    if (grid.y > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: bitmap too tall for one launch");
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), f, t);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

int ifhip_copy_rect(uint8_t* input, uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful, uint8_t* canvas,
                    uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride, int* canvas_alpha_meaningful, uint32_t from_x,
                    uint32_t from_y, uint32_t to_x, uint32_t to_y, uint32_t w, uint32_t h) {
    if (!canvas_alpha_meaningful) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null canvas_alpha_meaningful");
    Staged si, sc;
    int rc = si.up(input, in_w, in_h, in_stride, true);
    if (rc) return rc;
    if ((rc = sc.up(canvas, canvas_w, canvas_h, canvas_stride, true))) return rc;
    const int in_was = in_alpha_meaningful, cv_was = *canvas_alpha_meaningful;
    rc = ifhip_copy_rect_batch_device(si.d, si.bytes, in_w, in_h, in_stride, in_alpha_meaningful, sc.d, sc.bytes, canvas_w, canvas_h,
                                      canvas_stride, canvas_alpha_meaningful, from_x, from_y, to_x, to_y, w, h, 1, nullptr);
    if (rc) return rc;
    if (!in_was && (cv_was || *canvas_alpha_meaningful) && (rc = si.down(input))) return rc;      // the input was normalised
    return sc.down(canvas);
}

int ifhip_fill_rect(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int compositing, uint32_t x1, uint32_t y1,
                    uint32_t x2, uint32_t y2, uint32_t color_bgra) {
    Staged s;
    int rc = s.up(bgra, w, h, stride, true);
    if (rc) return rc;
    if ((rc = ifhip_fill_rect_batch_device(s.d, s.bytes, 1, w, h, stride, compositing, x1, y1, x2, y2, color_bgra, nullptr))) return rc;
    return s.down(bgra);
}

int ifhip_flip_vertical(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride) {
    Staged s;
    int rc = s.up(bgra, w, h, stride, true);
    if (rc) return rc;
    if ((rc = ifhip_flip_vertical_batch_device(s.d, s.bytes, 1, w, h, stride, nullptr))) return rc;
    return s.down(bgra);
}

int ifhip_flip_horizontal(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride) {
    Staged s;
    int rc = s.up(bgra, w, h, stride, true);
    if (rc) return rc;
    if ((rc = ifhip_flip_horizontal_batch_device(s.d, s.bytes, 1, w, h, stride, nullptr))) return rc;
    return s.down(bgra);
}

int ifhip_transpose(const uint8_t* from, uint32_t from_w, uint32_t from_h, uint32_t from_stride, uint8_t* to, uint32_t to_w,
                    uint32_t to_h, uint32_t to_stride) {
    if (from_w != to_h || from_h != to_w)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: For transposition, canvas and input formats must be the same and dimensions must be swapped");
    Staged sf, st;
    int rc = sf.up(from, from_w, from_h, from_stride, true);
    if (rc) return rc;
    if ((rc = st.up(to, to_w, to_h, to_stride, true))) return rc;          // row padding of the canvas is preserved
    if ((rc = ifhip_transpose_batch_device(sf.d, sf.bytes, from_w, from_h, from_stride, st.d, st.bytes, to_w, to_h, to_stride, 1, nullptr))) return rc;
    return st.down(to);
}

}  // extern "C"
