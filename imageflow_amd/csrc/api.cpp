// api.cpp -- the C ABI of libimageflow_hip.so (include/imageflow_hip.h) over the gfx950 kernels.
//
// Host responsibilities only: argument validation with the reference's error kinds (graphics/scaling.rs:24-48),
// per-shape plan construction (weights, vertical schedule, column strips), HBM staging for the host-buffer
// drop-in entry points, and launch geometry.  No pixel arithmetic happens on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "device.hpp"

namespace ifhip {
hipError_t launch_fused(const ResampleArgs& a, int slots, bool alpha, bool per_pixel, uint32_t grid, uint32_t block,
                        size_t lds, hipStream_t st);
hipError_t launch_generic(const ResampleArgs& a, bool alpha, float4* scratch, uint32_t img0, uint32_t n_img,
                          hipStream_t st);
hipError_t launch_banded(const ResampleArgs& a, bool alpha, const BandedArgs& b, uint32_t grid_x, size_t lds, hipStream_t st);
hipError_t launch_read_probe(const uint8_t* d, size_t bytes, uint32_t* sink, hipStream_t st);
hipError_t launch_mix_probe(const uint8_t* d, uint8_t* out, size_t bytes, uint32_t every, uint32_t* sink, hipStream_t st);
hipError_t launch_apply_matte(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                              uint32_t stride, uint32_t matte, float mb, float mg, float mr, float ma,
                              const float* s2l, const uint8_t* l2s, hipStream_t st);
}  // namespace ifhip

using namespace ifhip;

namespace {
// Scope guards: every early return of an entry point releases what it created.
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() {
        hipError_t e = hipEventCreate(&e0);
        return e != hipSuccess ? e : hipEventCreate(&e1);
    }
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
struct DeviceBuffer {
    void* p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
};
}  // namespace

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace {

constexpr size_t kLdsLimit = 160 * 1024;       // gfx950 LDS per CU == per-workgroup maximum
constexpr uint32_t kMaxStripOutputs = 2048;

// ---- per-device colour tables -----------------------------------------------------------------------
struct DeviceTables {
    float* s2l = nullptr;
    float* s2f = nullptr;
    uint8_t* l2s = nullptr;
    uint16_t* l2s_thr = nullptr;
};
std::mutex g_dev_mu;
std::map<int, DeviceTables> g_dev_tables;

int device_tables(DeviceTables* out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device (hipGetDevice failed); this library has no CPU path");
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_dev_tables.find(dev);
    if (it == g_dev_tables.end()) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        const ColorTables& t = color_tables();
        DeviceTables d;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.s2l), sizeof t.s2l));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.s2f), sizeof t.s2f));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.l2s), sizeof t.l2s));
        HIP_TRY(hipMemcpy(d.s2l, t.s2l, sizeof t.s2l, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.s2f, t.s2f, sizeof t.s2f, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.l2s, t.l2s, sizeof t.l2s, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.l2s_thr), sizeof t.l2s_thr));
        HIP_TRY(hipMemcpy(d.l2s_thr, t.l2s_thr, sizeof t.l2s_thr, hipMemcpyHostToDevice));
        it = g_dev_tables.emplace(dev, d).first;
    }
    *out = it->second;
    return IFHIP_OK;
}

template <typename T>
int upload(const std::vector<T>& v, T** out) {
    *out = nullptr;
    if (v.empty()) return IFHIP_OK;
    HIP_TRY(DEV_MALLOC(out, v.size() * sizeof(T)));
    HIP_TRY(static_cast<hipError_t>(copy_to_device(*out, v.data(), v.size() * sizeof(T))));
    return IFHIP_OK;
}

struct ScheduleOnDevice {
    VStep* steps = nullptr;
    uint32_t* band_begin = nullptr;
    uint32_t n_bands = 0;
};

}  // namespace

// ---- plan ------------------------------------------------------------------------------------------------
struct ifhip_resample_plan {
    int device = -1;
    uint32_t in_w = 0, in_h = 0, out_w = 0, out_h = 0;
    AxisWeights wv, wh;
    // device copies of the contribution tables
    uint32_t *d_v_left = nullptr, *d_v_count = nullptr, *d_v_off = nullptr;
    uint32_t *d_h_left = nullptr, *d_h_count = nullptr, *d_h_off = nullptr;
    float *d_v_w = nullptr, *d_h_w = nullptr, *d_h_wu = nullptr;
    uint4* d_h_meta = nullptr;
    // fast horizontal pass: every output runs the same number G <= 4 of 4-tap groups (moderate ratios)
    uint32_t h_fast_groups = 0;     // 0: not available (some output needs more than 4 groups)
    float* d_h_wg = nullptr;        // distinct weight rows, each zero-padded to G groups
    uint32_t h_wg_floats = 0;
    uint32_t* d_h_meta2 = nullptr;  // [out_w] first group | row id << 16
    // ... and its two-column form: G2 groups of 2 taps where that computes at most 2/3 of the taps per output (windows of 5-6 taps
    // aligned to 4 columns take 3 groups = 12 taps, aligned to 2 columns 4 groups = 8); no alpha, BGRA sources
    uint32_t h_two_groups = 0;      // 0: not available / not worth it
    float* d_h_wg2 = nullptr;       // distinct weight rows, each zero-padded to G2 groups of 2
    uint32_t h_wg2_floats = 0;
    uint32_t* d_h_meta3 = nullptr;  // [out_w] first 2-column group | row id << 16
    uint32_t h_wu_floats = 0;       // de-duplicated, 4-tap padded horizontal weight rows
    uint32_t h_avg_groups = 0;      // mean 4-tap groups per horizontal chain
    // fused-kernel geometry
    bool fused_possible = false;
    int slots = 0;
    struct StripSet {                // column strips for one (alpha) variant of the fused kernel
        std::vector<Strip> strips;
        Strip* d_strips = nullptr;
        uint32_t max_quads = 0;
        bool ok = false;
    } sets[2];                       // [in_alpha_meaningful]
    // lazily built, guarded by mu
    mutable std::mutex mu;
    mutable std::map<uint64_t, ScheduleOnDevice> schedules;     // key: bands | group << 32 | ahead << 40

    ~ifhip_resample_plan() {
        for (void* p : {(void*)d_v_left, (void*)d_v_count, (void*)d_v_off, (void*)d_h_left, (void*)d_h_count,
                        (void*)d_h_off, (void*)d_v_w, (void*)d_h_w, (void*)d_h_wu, (void*)d_h_meta, (void*)d_h_wg, (void*)d_h_meta2, (void*)d_h_wg2, (void*)d_h_meta3, (void*)sets[0].d_strips,
                        (void*)sets[1].d_strips})
            if (p) (void)DEV_FREE(p);
        for (auto& kv : schedules) {
            if (kv.second.steps) (void)DEV_FREE(kv.second.steps);
            if (kv.second.band_begin) (void)DEV_FREE(kv.second.band_begin);
        }
    }
};

namespace {

size_t fused_lds_bytes(uint32_t n_u, uint32_t nquads, int channels, uint32_t wu_floats, bool w_in_lds, bool l2s_in_lds,
                       uint32_t lut_copies_log2, bool per_pixel, uint32_t frames = 1, uint32_t fast_groups = 0) {
    return fused_lds_layout(n_u, nquads, wu_floats, channels, w_in_lds, l2s_in_lds, lut_copies_log2, per_pixel, frames, fast_groups).total;
}
// Horizontal pass mapping: one lane per output pixel (its C chains interleave, encode + store follow at once, no obuf
// round trip) measured faster than one lane per (pixel, channel) on every BASELINE shape (cfg2 -2.6 %, cfg2 with alpha
// -10 %, cfg3 -27 %); the per-channel form is kept for strips with less than one wave of outputs, where it is the only
// way to spread the (long) chains over more lanes.
bool use_per_pixel(uint32_t max_nu, int channels, uint32_t block) {
    return max_nu >= 64u || static_cast<uint64_t>(max_nu) * static_cast<uint32_t>(channels) > block;
}
uint32_t block_for(uint32_t max_quads, int px) {          // lanes of a frame slot: one per px source pixels, whole waves
    return std::max<uint32_t>(64u, (max_quads * static_cast<uint32_t>(4 / px) + 63u) & ~63u);
}
bool trace_launch() { return debug_switch("trace_launch") != nullptr; }      // one stderr line per launch: its geometry (tools/)
constexpr uint32_t kMinLutCopiesLog2 = 4;      // never fewer than 16 copies of the sRGB->float table (2-way conflicts)

// Split the output columns into strips whose staged source span fits one workgroup (max_lanes lanes x 4 px)
// and whose minimal LDS footprint fits the CU.
bool plan_strips(const AxisWeights& wh, uint32_t max_lanes, int px, int channels, std::vector<Strip>* out, uint32_t* max_quads) {
    for (uint32_t n = 1; n <= wh.n_out; ++n) {
        std::vector<Strip> s;
        bool ok = true;
        uint32_t mq = 0, mu = 0;
        for (uint32_t i = 0; i < n && ok; ++i) {
            Strip t;
            t.u0 = static_cast<uint32_t>(static_cast<uint64_t>(wh.n_out) * i / n);
            t.u1 = static_cast<uint32_t>(static_cast<uint64_t>(wh.n_out) * (i + 1) / n);
            if (t.u1 <= t.u0) { ok = false; break; }
            uint32_t lo = wh.left[t.u0], hi = 0;
            for (uint32_t u = t.u0; u < t.u1; ++u) {
                lo = std::min(lo, wh.left[u]);
                hi = std::max(hi, wh.left[u] + wh.count[u]);
            }
            t.cx0 = lo & ~3u;
            t.nquads = (hi - t.cx0 + 3u) / 4u;
            if (t.nquads > max_lanes || (t.u1 - t.u0) > kMaxStripOutputs) ok = false;
            mq = std::max(mq, t.nquads);
            mu = std::max(mu, t.u1 - t.u0);
            s.push_back(t);
        }
        if (ok) {
            const bool pp = use_per_pixel(mu, channels, block_for(mq, px));
            for (const Strip& t : s)
                if (fused_lds_bytes(t.u1 - t.u0, t.nquads, channels, 0, false, false, kMinLutCopiesLog2, pp) > kLdsLimit) ok = false;
        }
        if (ok) { *out = std::move(s); *max_quads = mq; return true; }
        if (n > 4096) break;
    }
    return false;
}

int get_schedule(const ifhip_resample_plan* p, uint32_t n_bands, int group, int ahead, ScheduleOnDevice* out) {
    std::lock_guard<std::mutex> lk(p->mu);
    const uint64_t key = static_cast<uint64_t>(n_bands) | (static_cast<uint64_t>(group) << 32) | (static_cast<uint64_t>(ahead) << 40);
    auto it = p->schedules.find(key);
    if (it == p->schedules.end()) {
        VSchedule s;
        if (!build_vschedule(p->wv, static_cast<int>(n_bands), group, ahead, &s))
            return fail(IFHIP_INVALID_STATE, "InvalidState: vertical schedule could not be built");
        ScheduleOnDevice d;
        d.n_bands = static_cast<uint32_t>(s.band_begin.size() - 1);
        int rc = upload(s.steps, &d.steps);
        if (rc) return rc;
        rc = upload(s.band_begin, &d.band_begin);
        if (rc) { (void)DEV_FREE(d.steps); return rc; }
        it = p->schedules.emplace(key, d).first;
    }
    *out = it->second;
    return IFHIP_OK;
}

std::atomic<uint32_t> g_cu_budget{0};            // ifhip_set_cu_budget: CUs the launches plan for (0: all of them)
constexpr uint32_t kComputeUnits = 256;          // MI355X

uint32_t choose_bands(const ifhip_resample_plan* p, uint32_t n_images, size_t n_strips) {
    // One workgroup occupies a CU (LDS), so a launch runs in ceil(workgroups / 256) rounds.  Cutting frames into bands
    // of output rows makes the rounds finer but every extra band re-reads its halo of source rows and stages the tables
    // again (a few microseconds per workgroup: `setup`, as a share of one frame's time on one CU); pick the band count with
    // the smallest estimated time.  Up to 64 bands: a launch of ONE frame (a job through the ABI) then spreads over 64 CUs
    // instead of 16 -- 3840x2160 -> 800x450 as a single frame: 118 us with 16 bands (round 5, profiles/r5_abi_*).
    const double wgs = static_cast<double>(n_images) * static_cast<double>(n_strips);
    const double halo = p->out_h ? static_cast<double>(p->wv.max_taps) / std::max<double>(1.0, p->in_h) : 0.0;
    const double setup = 0.01;
    const uint32_t budget = g_cu_budget.load(std::memory_order_relaxed);
    const double cus = budget ? static_cast<double>(budget) : static_cast<double>(kComputeUnits);
    const uint32_t max_bands = std::max<uint32_t>(1u, std::min<uint32_t>(64u, p->out_h / 4u));
    uint32_t best = 1;
    double best_cost = 1e300;
    for (uint32_t b = 1; b <= max_bands; ++b) {
        const double rounds = std::ceil(wgs * b / cus);
        const double cost = rounds * ((1.0 + halo * (b - 1)) / b + setup);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = b; }
    }
    return best;
}

bool fused_usable(const ifhip_resample_plan* p, int alpha, const uint8_t* d_in, size_t in_image_bytes, uint32_t in_stride,
                  const uint8_t* d_cb = nullptr, const uint8_t* d_cr = nullptr) {
    if (!p->fused_possible || !p->sets[alpha ? 1 : 0].ok) return false;
    if (d_cb) {                         // three component planes: 4-byte reads of 4 samples
        if (alpha || fused_shape(p->slots, 3).px != 4) return false;
        if (((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_cb) | reinterpret_cast<uintptr_t>(d_cr)) & 3u) ||
            (in_image_bytes & 3u) || (in_stride & 3u)) return false;
        for (const Strip& s : p->sets[0].strips)
            if (static_cast<uint64_t>(s.cx0) + 4u * s.nquads > in_stride) return false;
        return true;
    }
    if ((reinterpret_cast<uintptr_t>(d_in) & 15u) || (in_image_bytes & 15u) || (in_stride & 15u)) return false;
    for (const Strip& s : p->sets[alpha ? 1 : 0].strips)
        if (static_cast<uint64_t>(s.cx0 + 4u * s.nquads) * 4u > in_stride) return false;   // 16-byte row reads stay inside the row
    return true;
}

// Banded two-pass kernel (resample_kernels.hip): R output rows per workgroup, their source rows and vertically filtered
// rows in LDS beside the tables.  R is the largest of a short list for which two workgroups share a CU; failing that,
// whatever fits one.  A band's workgroups split the frames between them (frame_step), so that the tables are staged a few
// times per CU and not once per frame and band.
struct BandPlan { BandedArgs args{}; uint32_t grid = 0; size_t lds = 0; };
constexpr size_t kBandedTables = 16384 + 1024 + 16;
constexpr uint32_t kBandedWorkgroups = 8192;                 // sixteen rounds of two workgroups per CU (measured: 512 4.09, 1 024 3.91, 2 048 3.76, 4 096 3.70, 8 192 3.66 ms on the 3x shape)
bool banded_plan(const ifhip_resample_plan* p, const uint8_t* d_in, size_t in_image_bytes, uint32_t in_stride, uint32_t n_images, BandPlan* bp) {
    if ((reinterpret_cast<uintptr_t>(d_in) | in_image_bytes | in_stride) & 3u) return false;      // 4-byte pixel reads
    if (in_image_bytes > 0xffffffffull) return false;                                              // 32-bit offsets inside a frame
    const AxisWeights& wv = p->wv;
    const uint32_t out_h = p->out_h;
    auto src_rows_of = [&](uint32_t R) {                    // widest source window of any band of R output rows
        uint32_t worst = 0;
        for (uint32_t j0 = 0; j0 < out_h; j0 += R) {
            uint32_t lo = 0xffffffffu, hi = 0;
            for (uint32_t j = j0; j < std::min(out_h, j0 + R); ++j) { lo = std::min(lo, wv.left[j]); hi = std::max(hi, wv.left[j] + wv.count[j]); }
            worst = std::max(worst, hi - lo);
        }
        return worst;
    };
    uint32_t src_rows_memo[65] = {};                        // (asked for the same dozen R by every candidate strip width)
    auto src_rows = [&](uint32_t R) { return R <= 64u ? (src_rows_memo[R] ? src_rows_memo[R] : (src_rows_memo[R] = src_rows_of(R))) : src_rows_of(R); };
    bool ascending = true;                                  // window starts and ends never step back (they do not, but the kernel's
    for (uint32_t j = 1; j < out_h; ++j)                    // shortcut rests on it, so it is checked, not assumed)
        if (wv.left[j] < wv.left[j - 1] || wv.left[j] + wv.count[j] < wv.left[j - 1] + wv.count[j - 1]) ascending = false;
    const AxisWeights& wh = p->wh;
    const uint32_t out_w = p->out_w;
    static const uint32_t kRows[] = {64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2, 1};
    uint32_t wgs = kBandedWorkgroups;
    if (const char* e = debug_switch("banded_wgs")) wgs = static_cast<uint32_t>(std::max(1, std::atoi(e)));   // test hook: the frame loop of a workgroup
    auto commit = [&](uint32_t R, uint32_t ns, uint32_t strip_w, uint32_t hwf, bool h_lds, size_t lds) {
        BandedArgs& b = bp->args;
        b.rows_per_band = R; b.n_bands = (out_h + R - 1u) / R; b.src_rows_cap = ns;
        b.strip_w = strip_w; b.n_strips = (out_w + strip_w - 1u) / strip_w;
        b.frame_step = std::max<uint32_t>(1u, std::min<uint32_t>(n_images, wgs / std::max(1u, b.n_bands * b.n_strips)));
        b.h_w_floats = hwf;
        b.flags = (ascending ? 2u : 0u) | (h_lds ? 4u : 0u);
        bp->grid = b.n_bands * b.n_strips * b.frame_step;
        bp->lds = lds;
    };
    // ---- whole rows (small frames): R is the largest of the list for which two workgroups share a CU, else whatever fits one ----
    // horizontal tables in LDS when they are small (up-scales: ~5 taps per output column)
    const size_t h_bytes = ((3u * static_cast<size_t>(out_w) + wh.w.size()) * 4u + 15u) & ~static_cast<size_t>(15u);
    const bool h_lds = h_bytes <= 32u * 1024u;
    const size_t tables = kBandedTables + (h_lds ? h_bytes : 0u);
    const size_t row_bytes = static_cast<size_t>(p->in_w) * 16u;
    uint32_t whole_R = 0, whole_ns = 0; size_t whole_lds = 0;
    for (int pass = 0; pass < 2 && !whole_R; ++pass) {
        const size_t budget = pass == 0 ? kLdsLimit / 2 : kLdsLimit;
        for (uint32_t R0 : kRows) {
            const uint32_t R = std::min(R0, out_h);
            if (pass == 0 && R < 4u && out_h >= 4u) break;
            const uint32_t ns = src_rows(R);
            const size_t lds = tables + static_cast<size_t>(ns + R) * row_bytes;
            if (lds <= budget) { whole_R = R; whole_ns = ns; whole_lds = lds; break; }
        }
    }
    // ---- column strips (wide frames): where whole rows leave a band of fewer than 16 rows (each band converts its own halo of
    // source rows and stages the tables again) or do not fit at all, a workgroup takes a strip of S output columns of a band
    // of R rows; its source columns and its slice of the weights are the union of its columns' windows. ----
    uint32_t forced_strip = 0;
    if (const char* e = debug_switch("banded_strip")) forced_strip = static_cast<uint32_t>(std::max(0, std::atoi(e)));   // test hook: strips on small frames
    const bool whole_good = whole_R != 0 && (whole_R >= 16u || whole_R >= out_h);
    if ((whole_good && !forced_strip) || out_w < 2u) {
        if (!whole_R) return false;
        commit(whole_R, whole_ns, out_w, static_cast<uint32_t>(wh.w.size()), h_lds, whole_lds);
        return true;
    }
    auto pad64 = [](uint32_t v) { return (v + 63u) / 64u * 64u; };
    const double nv = static_cast<double>(wv.w.size()) / std::max(1u, out_h);          // mean taps of the vertical windows
    // cost per output pixel, in tap steps (one 16-byte LDS read + its multiply-adds): converting the tile's source pixels (three
    // table reads each: 3), the vertical pass over the strip's source columns, both with their idle lanes; a tile that leaves no
    // room for a second workgroup on the CU waits out its own barriers (x 1.25)
    auto cost_of = [&](uint32_t R, uint32_t ns, uint32_t S, uint32_t sc, size_t lds) {
        const double px = static_cast<double>(R) * S;
        return (3.0 * ns * pad64(sc) + nv * R * pad64(sc) + 8.0 * R * pad64(S)) / px * (lds > kLdsLimit / 2 ? 1.25 : 1.0);
    };
    double best = 1e300;
    struct { uint32_t R, ns, S, hwf; bool h_lds; size_t lds; } pick{};
    if (whole_R && !forced_strip) {
        best = cost_of(whole_R, whole_ns, out_w, p->in_w, whole_lds);
        pick = {whole_R, whole_ns, out_w, static_cast<uint32_t>(wh.w.size()), h_lds, whole_lds};
    }
    static const uint32_t kStrips[] = {512, 384, 256, 192, 128, 112, 96, 64, 48, 32, 16};
    for (uint32_t S0 : kStrips) {
        const uint32_t S = forced_strip ? std::min(forced_strip, out_w) : S0;
        if (S >= out_w && !forced_strip) continue;
        uint32_t sc = 0, hwf = 0;                            // widest strip: source columns, floats of its weight slice
        for (uint32_t u0 = 0; u0 < out_w; u0 += S) {
            const uint32_t u1 = std::min(out_w, u0 + S);
            uint32_t lo = 0xffffffffu, hi = 0, wlo = 0xffffffffu, whi = 0;       // (as the kernel finds them)
            for (uint32_t u = u0; u < u1; ++u) {
                lo = std::min(lo, wh.left[u]); hi = std::max(hi, wh.left[u] + wh.count[u]);
                wlo = std::min(wlo, wh.offset[u]); whi = std::max(whi, wh.offset[u] + wh.count[u]);
            }
            sc = std::max(sc, hi - lo);
            hwf = std::max(hwf, whi - wlo);
        }
        const size_t hb = ((3u * static_cast<size_t>(S) + hwf) * 4u + 15u) & ~static_cast<size_t>(15u);
        const bool hl = hb <= 32u * 1024u;
        for (uint32_t R0 : kRows) {
            const uint32_t R = std::min(R0, out_h);
            const uint32_t ns = src_rows(R);
            const size_t lds = kBandedTables + (hl ? hb : 0u) + static_cast<size_t>(ns + R) * sc * 16u;
            if (lds > kLdsLimit) continue;
            const double c = cost_of(R, ns, S, sc, lds);
            if (c < best) { best = c; pick = {R, ns, S, hwf, hl, lds}; }
        }
        if (forced_strip) break;
    }
    if (best == 1e300) return false;
    commit(pick.R, pick.ns, pick.S, pick.hwf, pick.h_lds, pick.lds);
    return true;
}
// The banded kernel stands in for the generic pair wherever it fits (measured, MI355X: 3x up-scale 9.97 -> 3.65 ms), never for
// the fused kernel (the 2x up-scale the fused kernel takes is faster there: 2.93 vs 4.43 ms).

int validate_render(uint32_t in_w, uint32_t in_h, uint32_t in_stride, uint32_t cw, uint32_t ch, uint32_t c_stride,
                    uint32_t x, uint32_t y, uint32_t w, uint32_t h, int working_space, int compositing, uint32_t in_px_bytes = 4) {
    if (static_cast<uint64_t>(h) + y > ch || static_cast<uint64_t>(w) + x > cw)                 // scaling.rs:24-29
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Destination rectangle for scale2d is out of bounds");
    if (w == 0 || h == 0 || in_w == 0 || in_h == 0)                                              // bitmaps.rs:700-702
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Bitmap dimensions cannot be zero");
    if (static_cast<uint64_t>(in_w) * in_px_bytes > in_stride || static_cast<uint64_t>(cw) * 4u > c_stride)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: stride smaller than a BGRA row");
    if (c_stride & 3u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: canvas stride must be a multiple of 4 bytes");
    if (working_space != IFHIP_SPACE_SRGB && working_space != IFHIP_SPACE_LINEAR)
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: working floatspace %d", working_space);
    if (compositing < IFHIP_REPLACE_SELF || compositing > IFHIP_BLEND_WITH_MATTE)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: compositing mode %d", compositing);
    return IFHIP_OK;
}

int enqueue_batch(const ifhip_resample_plan* p, const uint8_t* d_in, size_t in_image_bytes, uint32_t in_stride,
                  int alpha, uint32_t n_images, uint8_t* d_canvas, size_t canvas_image_bytes, uint32_t cw, uint32_t ch,
                  uint32_t c_stride, uint32_t x, uint32_t y, int working_space, int compositing, uint32_t matte,
                  float* d_f32, int force_kernel, hipStream_t st, const uint8_t* d_cb = nullptr, const uint8_t* d_cr = nullptr,
                  bool probe = false) {
    // probe (planar source only): everything up to the launch -- IFHIP_OK means the real call will run the fused kernel
    // d_cb / d_cr: planar YCbCr source (d_in = the Y plane, in_stride = sample pitch, in_image_bytes = plane size).  That form
    // exists only on the fused kernel: kNotFusable tells the caller to go through a BGRA bitmap instead.
    if (!p) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null plan");
    const bool ycc = d_cb != nullptr;
    int rc = validate_render(p->in_w, p->in_h, in_stride, cw, ch, c_stride, x, y, p->out_w, p->out_h, working_space, compositing, ycc ? 1u : 4u);
    if (rc) return rc;
    if (n_images == 0) return IFHIP_OK;
    if (!d_in || !d_canvas || (ycc && !d_cr)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null bitmap pointer");
    if ((reinterpret_cast<uintptr_t>(d_canvas) & 3u) || (canvas_image_bytes & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: canvas pixels must be 4-byte aligned");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != p->device) return fail(IFHIP_INVALID_STATE, "InvalidState: plan belongs to device %d, current device is %d", p->device, dev);
    DeviceTables tb;
    rc = device_tables(&tb);
    if (rc) return rc;
    const ColorTables& host_tb = color_tables();

    ResampleArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = d_in; a.in_image_bytes = in_image_bytes; a.in_stride = in_stride; a.in_w = p->in_w; a.in_h = p->in_h;
    a.in_cb = d_cb; a.in_cr = d_cr; a.ycc = ycc ? 1u : 0u;
    a.canvas = d_canvas; a.canvas_image_bytes = canvas_image_bytes; a.c_stride = c_stride; a.x = x; a.y = y;
    a.out_w = p->out_w; a.out_h = p->out_h; a.f32_dump = d_f32;
    a.h_meta = p->d_h_meta; a.h_wu = p->d_h_wu; a.h_wu_floats = p->h_wu_floats;
    a.h_left = p->d_h_left; a.h_count = p->d_h_count;
    a.v_left = p->d_v_left; a.v_count = p->d_v_count; a.v_off = p->d_v_off; a.v_w = p->d_v_w;
    a.h_off = p->d_h_off; a.h_w = p->d_h_w;
    a.linear = working_space == IFHIP_SPACE_LINEAR;
    a.lut_in = a.linear ? tb.s2l : tb.s2f;
    a.l2s = tb.l2s;
    a.l2s_thr = tb.l2s_thr;
    a.mode = compositing;
    const float* s2 = a.linear ? host_tb.s2l : host_tb.s2f;       // matte colour in working space, scaling.rs:141-143
    a.m0 = s2[matte & 255u]; a.m1 = s2[(matte >> 8) & 255u]; a.m2 = s2[(matte >> 16) & 255u];
    a.matte_a = static_cast<float>(matte >> 24) * (1.0f / 255.0f);
    a.n_images = n_images;

    bool fused = fused_usable(p, alpha, d_in, in_image_bytes, in_stride, d_cb, d_cr);
    if (ycc && !fused) return kNotFusable;
    if (force_kernel == 0 && !fused)
        return fail(IFHIP_INVALID_STATE, "InvalidState: fused kernel requested but its preconditions do not hold "
                    "(live rows %d > %d, or rows not 16-byte aligned / padded)", p->slots, kMaxSlots);
    if (force_kernel == 1) fused = false;

    // banded two-pass kernel: asked for (force_kernel 2), or in auto mode where the fused kernel does not apply
    if (!ycc && (force_kernel == 2 || force_kernel == -1)) {
        const bool want = force_kernel == 2 || !fused;
        BandPlan bp;
        if (want && banded_plan(p, d_in, in_image_bytes, in_stride, n_images, &bp)) {
            // test hook: masks the plan's flags (2 band rows from its first and last row, 4 horizontal tables in LDS) so that the
            // kernel's table-free forms, which real weight tables reach only at very wide outputs, run in the suite
            if (const char* fe = debug_switch("banded_flags")) bp.args.flags &= static_cast<uint32_t>(std::atoi(fe));
            if (trace_launch())
                std::fprintf(stderr, "ifhip banded launch: %ux%u -> %ux%u alpha=%d rows/band=%u bands=%u src rows=%u strip=%u strips=%u frame step=%u flags=%u grid=%u lds=%zu images=%u\n",
                             p->in_w, p->in_h, p->out_w, p->out_h, alpha, bp.args.rows_per_band, bp.args.n_bands, bp.args.src_rows_cap,
                             bp.args.strip_w, bp.args.n_strips, bp.args.frame_step, bp.args.flags, bp.grid, bp.lds, n_images);
            HIP_TRY(launch_banded(a, alpha != 0, bp.args, bp.grid, bp.lds, st));
            return IFHIP_OK;
        }
        if (force_kernel == 2)
            return fail(IFHIP_INVALID_STATE, "InvalidState: banded kernel requested but a band's source rows do not fit the LDS "
                        "(or the source pixels are not 4-byte aligned)");
    }

    if (fused) {
        const ifhip_resample_plan::StripSet& ss = p->sets[alpha ? 1 : 0];
        a.strips = ss.d_strips; a.n_strips = static_cast<uint32_t>(ss.strips.size());
        const int channels = alpha ? 4 : 3;
        const uint32_t block = block_for(ss.max_quads, fused_shape(p->slots, channels).px);
        uint32_t max_nu = 0;
        for (const Strip& s : ss.strips) max_nu = std::max(max_nu, s.u1 - s.u0);
        const bool per_pixel = use_per_pixel(max_nu, channels, block);
        const size_t limit = kLdsLimit;
        // The fast horizontal pass (same group count G for every output, rows padded with +0 weights) needs the padded
        // weight rows in LDS and the per-pixel mapping; when that does not fit, plan again for the general pass.
        uint32_t frames = 1, copies_log2 = kMinLutCopiesLog2, fast_g = 0, wu_floats = p->h_wu_floats;
        bool w_in_lds = false, l2s_in_lds = false;
        ScheduleOnDevice sd;
        // forms of the horizontal pass, best first: two-column groups, four-column groups (both: the fast pass), general
        const bool fast_ok = per_pixel && p->h_fast_groups && fused_shape(p->slots, channels).px == 4;
        const bool two_ok = fast_ok && p->h_two_groups && !alpha && !ycc;
        bool two = false;
        for (int attempt = two_ok ? -1 : (fast_ok ? 0 : 1); attempt < 2; ++attempt) {
            two = attempt < 0;
            fast_g = two ? p->h_two_groups : (attempt == 0 ? p->h_fast_groups : 0u);
            wu_floats = two ? p->h_wg2_floats : (fast_g ? p->h_wg_floats : p->h_wu_floats);
            // Frames per workgroup: a source narrower than half the workgroup would leave the CU with a handful of waves
            // (one workgroup per CU: the tables fill most of the LDS), so F frames share a workgroup and its tables.
            frames = 1;
            if (ss.strips.size() == 1) {
                const uint32_t max_f = std::min<uint32_t>(static_cast<uint32_t>(fused_max_threads(p->slots, channels)) / block, n_images);
                const Strip& s0 = ss.strips[0];
                for (uint32_t f = max_f; f > 1; --f)
                    if (fused_lds_bytes(s0.u1 - s0.u0, s0.nquads, channels, wu_floats, true, a.linear != 0, kMinLutCopiesLog2, per_pixel, f, fast_g) <= limit) {
                        frames = f;
                        break;
                    }
            }
            // bands: by the number of workgroups the launch really has (frames / F per strip)
            const uint32_t want_bands = choose_bands(p, (n_images + frames - 1u) / frames, ss.strips.size());
            rc = get_schedule(p, want_bands, fused_shape(p->slots, channels).rows_in_flight, fused_lookahead(p->slots, channels), &sd);
            if (rc) return rc;
            a.steps = sd.steps; a.band_begin = sd.band_begin; a.n_bands = sd.n_bands;
            // LDS budget beyond the minimum the strips were planned for (double-buffered rows + 16 copies of the sRGB->float
            // table): the de-duplicated horizontal weight rows, then -- by lookup cost -- the second 16 table copies and the
            // 16 KiB linear->sRGB table (otherwise encoded by threshold search)
            auto fits = [&](bool w, bool l2s, uint32_t copies_log2) {
                for (const Strip& s : ss.strips)
                    if (fused_lds_bytes(s.u1 - s.u0, s.nquads, channels, wu_floats, w, l2s, copies_log2, per_pixel, frames, fast_g) > limit) return false;
                return true;
            };
            w_in_lds = fits(true, false, kMinLutCopiesLog2);
            // What goes next depends on where the lookups are: the 16 KiB linear->sRGB table saves an 8-step threshold
            // search (~40 instructions) per encoded channel, the second set of 16 table copies saves one LDS conflict cycle
            // per converted sample.  Per output row a strip encodes 3*n_u channels and converts 12*nquads*(in_h/out_h)
            // samples; thumbnail-sized outputs (cfg2) want the copies first, moderate ratios (cfg3) the encode table.
            const double enc_cost = 3.0 * max_nu * 40.0;
            const double conv_cost = 12.0 * ss.max_quads * (static_cast<double>(p->in_h) / std::max<uint32_t>(1u, p->out_h)) * 2.0;
            const bool l2s_allowed = a.linear != 0;
            copies_log2 = kMinLutCopiesLog2;
            l2s_in_lds = false;
            if (enc_cost > conv_cost) {
                l2s_in_lds = l2s_allowed && fits(w_in_lds, true, kMinLutCopiesLog2);
                if (fits(w_in_lds, l2s_in_lds, 5)) copies_log2 = 5u;
            } else {
                if (fits(w_in_lds, false, 5)) copies_log2 = 5u;
                l2s_in_lds = l2s_allowed && fits(w_in_lds, true, copies_log2);
            }
            if (!fast_g || w_in_lds) break;
        }
        if (ycc && !(w_in_lds && per_pixel)) return kNotFusable;     // the planar source is instantiated for that form only
        a.h_groups = two ? 16u + fast_g : fast_g;                    // (16 + G2: the two-column form)
        if (two) { a.h_wu = p->d_h_wg2; a.h_wu_floats = p->h_wg2_floats; a.h_meta2 = p->d_h_meta3; }
        else if (fast_g) { a.h_wu = p->d_h_wg; a.h_wu_floats = p->h_wg_floats; a.h_meta2 = p->d_h_meta2; }
        a.lut_copies_log2 = copies_log2;
        a.h_w_in_lds = w_in_lds ? 1u : 0u;
        a.l2s_in_lds = l2s_in_lds ? 1u : 0u;
        size_t lds = 0;
        for (const Strip& s : ss.strips)
            lds = std::max(lds, fused_lds_bytes(s.u1 - s.u0, s.nquads, channels, wu_floats, w_in_lds, l2s_in_lds, copies_log2, per_pixel, frames, fast_g));
        if (lds > kLdsLimit) return fail(IFHIP_INVALID_STATE, "InvalidState: fused kernel LDS plan exceeds the CU (%zu bytes)", lds);
        a.frames_per_wg = frames;
        a.lanes_per_frame = block;
        const uint64_t grid = static_cast<uint64_t>((n_images + frames - 1u) / frames) * sd.n_bands * a.n_strips;
        if (grid > 0x7fffffffull) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: batch too large for one launch");
        // Up-scales with alpha whose geometry leaves the CU a workgroup of at most four waves (a 1 440 - 2 048 column source cut
        // into two strips, the output rows of one frame filling the LDS so that no second frame shares the workgroup) run on the
        // banded kernel's column strips instead: 0.34 - 0.89 of the fused kernel's time on every such shape and filter measured
        // (profiles/r6_fused_vs_banded_upscales.jsonl); with five or more waves, and without alpha, the fused kernel stays ahead.
        if (force_kernel == -1 && !ycc && !probe && alpha && block * frames <= 256u &&
            4ull * p->out_w >= 5ull * p->in_w && 4ull * p->out_h >= 5ull * p->in_h) {
            BandPlan bp;
            if (banded_plan(p, d_in, in_image_bytes, in_stride, n_images, &bp)) {
                if (trace_launch())
                    std::fprintf(stderr, "ifhip banded launch (instead of a %u-lane fused workgroup): %ux%u -> %ux%u rows/band=%u strip=%u strips=%u grid=%u lds=%zu images=%u\n",
                                 block * frames, p->in_w, p->in_h, p->out_w, p->out_h, bp.args.rows_per_band, bp.args.strip_w, bp.args.n_strips, bp.grid, bp.lds, n_images);
                HIP_TRY(launch_banded(a, true, bp.args, bp.grid, bp.lds, st));
                return IFHIP_OK;
            }
        }
        if (trace_launch())                                      // development aid: the shape this call launches
            std::fprintf(stderr, "ifhip fused launch: %ux%u -> %ux%u K=%d alpha=%d ycc=%d lanes/frame=%u frames/wg=%u bands=%u strips=%u "
                         "grid=%llu lds=%zu fast_g=%u two_col=%d w_in_lds=%d l2s_in_lds=%d lut_copies=%u per_pixel=%d images=%u\n",
                         p->in_w, p->in_h, p->out_w, p->out_h, p->slots, alpha, ycc ? 1 : 0, block, frames, sd.n_bands, a.n_strips,
                         static_cast<unsigned long long>(grid), lds, fast_g, two ? 1 : 0, w_in_lds ? 1 : 0, l2s_in_lds ? 1 : 0, 1u << copies_log2,
                         per_pixel ? 1 : 0, n_images);
        if (probe) return IFHIP_OK;
        HIP_TRY(launch_fused(a, p->slots, alpha != 0, per_pixel, static_cast<uint32_t>(grid), block * frames, lds, st));
        return IFHIP_OK;
    }

    // generic two-pass path through an HBM scratch of [chunk][out_h][in_w] float4
    if (p->out_h > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: output taller than 65535 rows");
    const size_t per_image = static_cast<size_t>(p->out_h) * p->in_w * sizeof(float4);
    const size_t budget = static_cast<size_t>(1) << 30;
    uint32_t chunk = static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(n_images, budget / std::max<size_t>(per_image, 1))));
    chunk = std::min<uint32_t>(chunk, 65535u);
    // stream-ordered scratch from the block cache: nothing is shared between concurrent calls on the same plan, and the
    // memory is reusable as soon as the last kernel of this call has run (no host wait here)
    float4* scratch = nullptr;
    HIP_TRY(static_cast<hipError_t>(cached_malloc_for_stream(reinterpret_cast<void**>(&scratch), per_image * chunk, st, true)));
    hipError_t le = hipSuccess;
    for (uint32_t i0 = 0; i0 < n_images && le == hipSuccess; i0 += chunk) {
        const uint32_t n = std::min(chunk, n_images - i0);
        le = launch_generic(a, alpha != 0, scratch, i0, n, st);
    }
    const hipError_t fe = static_cast<hipError_t>(cached_free_after(scratch, st));
    HIP_TRY(le);
    HIP_TRY(fe);
    return IFHIP_OK;
}

}  // namespace

namespace ifhip {
// The resampler fed by the JPEG stage's component planes (jpeg_kernels.hip).  IFHIP_OK, an error, or kNotFusable.
int resample_from_ycc_planes_v(const ifhip_resample_plan* plan, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr,
                             size_t plane_bytes, uint32_t pitch, uint32_t n_images, uint8_t* d_canvas, size_t canvas_image_bytes,
                             uint32_t cw, uint32_t ch, uint32_t c_stride, uint32_t x, uint32_t y, int working_space, int compositing,
                             uint32_t matte, void* hip_stream, bool probe) {
    return enqueue_batch(plan, d_y, plane_bytes, pitch, 0, n_images, d_canvas, canvas_image_bytes, cw, ch, c_stride, x, y,
                         working_space, compositing, matte, nullptr, -1, static_cast<hipStream_t>(hip_stream), d_cb, d_cr, probe);
}
void resample_plan_shape(const ifhip_resample_plan* plan, uint32_t* in_w, uint32_t* in_h, uint32_t* out_w, uint32_t* out_h) {
    *in_w = plan->in_w; *in_h = plan->in_h; *out_w = plan->out_w; *out_h = plan->out_h;
}
}  // namespace ifhip

// ======================================================================================================
// extern "C"
// ======================================================================================================
extern "C" {

const char* ifhip_last_error_message(void) { return last_error(); }
const char* ifhip_version(void) { return "imageflow_hip 0.1 (gfx950)"; }

int ifhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
    return usable;
}

int ifhip_set_cu_budget(uint32_t compute_units) {
    if (compute_units > kComputeUnits) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: CU budget %u (the device has %u)", compute_units, kComputeUnits);
    g_cu_budget.store(compute_units, std::memory_order_relaxed);
    return IFHIP_OK;
}

int ifhip_set_device(int ordinal) {
    if (hipSetDevice(ordinal) != hipSuccess)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: hipSetDevice(%d) failed", ordinal);
    return IFHIP_OK;
}

uint32_t ifhip_stride_for_width(uint32_t w) {
    const uint64_t row = (static_cast<uint64_t>(w) * 4u + 63u) / 64u * 64u;
    return row > 0xFFFFFFC0ull ? 0u : static_cast<uint32_t>(row);          // 0 = the width has no 32-bit stride
}

int ifhip_populate_weights(int filter, int lobe_mode, float lobe_value, double kernel_width_scale,
                           uint32_t output_line_size, uint32_t input_line_size, uint32_t* left_pixel,
                           uint32_t* tap_count, float* weights, uint32_t weights_capacity, uint32_t* n_weights) {
    FilterSpec spec;
    if (!filter_spec_for(filter, &spec)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: unknown filter %d", filter);
    spec.blur *= kernel_width_scale;                 // set_kernel_width_scale, weights.rs:155-157
    spec.lobe_mode = lobe_mode; spec.lobe_value = lobe_value;
    AxisWeights w;
    int rc = build_axis_weights(spec, output_line_size, input_line_size, &w);
    if (rc) return rc;
    if (n_weights) *n_weights = static_cast<uint32_t>(w.w.size());
    if (weights_capacity == 0) return IFHIP_OK;
    if (w.w.size() > weights_capacity) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: weights_capacity too small");
    if (left_pixel) std::memcpy(left_pixel, w.left.data(), w.left.size() * sizeof(uint32_t));
    if (tap_count) std::memcpy(tap_count, w.count.data(), w.count.size() * sizeof(uint32_t));
    if (weights) std::memcpy(weights, w.w.data(), w.w.size() * sizeof(float));
    return IFHIP_OK;
}

int ifhip_table_srgb_to_floatspace(int working_space, float* out256) {
    if (!out256) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null table pointer");
    const ColorTables& t = color_tables();
    if (working_space == IFHIP_SPACE_LINEAR) std::memcpy(out256, t.s2l, sizeof t.s2l);
    else if (working_space == IFHIP_SPACE_SRGB) std::memcpy(out256, t.s2f, sizeof t.s2f);
    else return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: working floatspace %d", working_space);
    return IFHIP_OK;
}

int ifhip_table_linear_to_srgb(uint8_t* out16384) {
    if (!out16384) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null table pointer");
    std::memcpy(out16384, color_tables().l2s, 16384);
    return IFHIP_OK;
}

int ifhip_table_linear_to_srgb_thresholds(uint16_t* out256) {
    if (!out256) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null table pointer");
    std::memcpy(out256, color_tables().l2s_thr, 512);
    return IFHIP_OK;
}

int ifhip_resample_plan_create(ifhip_resample_plan** plan, uint32_t in_w, uint32_t in_h, uint32_t w, uint32_t h,
                               int filter, float sharpen_percent_goal) {
    if (!plan) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null plan out-pointer");
    *plan = nullptr;
    if (w == 0 || h == 0 || in_w == 0 || in_h == 0)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: Bitmap dimensions cannot be zero");
    FilterSpec spec;
    if (!filter_spec_for(filter, &spec)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: unknown filter %d", filter);
    if (sharpen_percent_goal > 0.0f) {               // scaling.rs:103-105 -> LobeRatio::SharpenPercent
        spec.lobe_mode = IFHIP_LOBE_SHARPEN_PERCENT;
        spec.lobe_value = sharpen_percent_goal;
    }
    DeviceTables tb;
    int rc = device_tables(&tb);
    if (rc) return rc;
    std::unique_ptr<ifhip_resample_plan> p(new ifhip_resample_plan);
    HIP_TRY(hipGetDevice(&p->device));
    p->in_w = in_w; p->in_h = in_h; p->out_w = w; p->out_h = h;
    rc = build_axis_weights(spec, h, in_h, &p->wv);
    if (rc) return rc;
    rc = build_axis_weights(spec, w, in_w, &p->wh);
    if (rc) return rc;

    // Horizontal weight rows for the fused kernel.  A row starts at the output's first tap rounded DOWN to a multiple
    // of 4 source columns (so that a lane gathers 4 taps with one aligned 16-byte LDS read), the skipped columns get
    // weight +0.0f (exact: fmaf(+0, x, +0) == +0 for finite x, and the chain starts at +0), and the row is zero-padded
    // to a multiple of 4 (the kernel predicates the taps of the last group).  Rows are then de-duplicated bit for
    // bit: at rational scale factors they recur with period out_w / gcd(in_w, out_w) (3840 -> 200: 36 rows of 200),
    // which is what lets the whole table live in LDS.
    std::vector<float> wu;
    std::vector<uint4> hmeta(w);
    {
        std::map<std::vector<uint32_t>, uint32_t> seen;        // row bits -> offset in wu
        uint64_t groups = 0;
        for (uint32_t u = 0; u < w; ++u) {
            const uint32_t n = p->wh.count[u], lead = p->wh.left[u] & 3u, total = lead + n, npad = (total + 3u) & ~3u;
            std::vector<uint32_t> bits(npad, 0u);
            std::memcpy(bits.data() + lead, p->wh.w.data() + p->wh.offset[u], n * sizeof(float));
            auto it = seen.find(bits);
            if (it == seen.end()) {
                const uint32_t off = static_cast<uint32_t>(wu.size());
                wu.resize(wu.size() + npad, 0.0f);
                std::memcpy(wu.data() + off, bits.data(), npad * sizeof(float));
                it = seen.emplace(std::move(bits), off).first;
            }
            hmeta[u] = make_uint4(p->wh.left[u] & ~3u, npad / 4u, it->second, ((total - 1u) & 3u) + 1u);
            groups += npad / 4u;
        }
        p->h_wu_floats = static_cast<uint32_t>(wu.size());
        p->h_avg_groups = static_cast<uint32_t>((groups + w - 1u) / w);
    }
    // Fast horizontal pass: when no output needs more than 4 groups, pad every row to the common count G -- the extra
    // groups carry weight +0 (exact, as the leading zeros above) -- so that the kernel runs G unrolled groups per
    // output with immediate LDS offsets, no per-lane trip count and a 4-byte record per output.
    std::vector<float> wg;
    std::vector<uint32_t> hmeta2(w);
    {
        uint32_t g_max = 0;
        for (uint32_t u = 0; u < w; ++u) g_max = std::max(g_max, hmeta[u].y);
        if (g_max >= 2u && g_max <= 4u) {
            std::map<std::vector<uint32_t>, uint32_t> seen;        // padded row bits -> row id
            const uint32_t row_floats = g_max * 4u;
            bool ok = true;
            for (uint32_t u = 0; u < w && ok; ++u) {
                std::vector<uint32_t> bits(row_floats, 0u);
                std::memcpy(bits.data(), wu.data() + hmeta[u].z, hmeta[u].y * 16u);
                auto it = seen.find(bits);
                if (it == seen.end()) {
                    const uint32_t id = static_cast<uint32_t>(seen.size());
                    if (id >= 65536u) { ok = false; break; }
                    wg.resize(wg.size() + row_floats);
                    std::memcpy(wg.data() + static_cast<size_t>(id) * row_floats, bits.data(), row_floats * 4u);
                    it = seen.emplace(std::move(bits), id).first;
                }
                if ((hmeta[u].x >> 2) >= 65536u) { ok = false; break; }
                hmeta2[u] = (hmeta[u].x >> 2) | (it->second << 16);
            }
            if (ok) { p->h_fast_groups = g_max; p->h_wg_floats = static_cast<uint32_t>(wg.size()); }
        }
    }
    // The same with groups of TWO source columns (8-byte LDS reads): a row starts at the first tap rounded down to an even
    // column.  Worth it where it computes a third fewer taps per output: 1600 -> 1200 Robidoux has 5-6 taps per output,
    // 3 groups of 4 (12 taps) or 4 groups of 2 (8): cfg3 level 1 2.91 -> 2.70 ms; 1200 -> 400 (16 taps or 12) measured equal
    // and stays with groups of four.  The taps keep their order and the padding is +0: same pixels.
    std::vector<float> wg2;
    std::vector<uint32_t> hmeta3(w);
    if (p->h_fast_groups) {
        uint32_t g2_max = 0;
        for (uint32_t u = 0; u < w; ++u) g2_max = std::max(g2_max, ((p->wh.left[u] & 1u) + p->wh.count[u] + 1u) >> 1);
        if (g2_max >= 2u && g2_max <= 6u && 3u * g2_max <= 4u * p->h_fast_groups) {     // at most 2/3 of the taps (measured: 3/4 gains nothing)
            std::map<std::vector<uint32_t>, uint32_t> seen;
            const uint32_t row_floats = g2_max * 2u;
            bool ok = true;
            for (uint32_t u = 0; u < w && ok; ++u) {
                std::vector<uint32_t> bits(row_floats, 0u);
                std::memcpy(bits.data() + (p->wh.left[u] & 1u), p->wh.w.data() + p->wh.offset[u], p->wh.count[u] * sizeof(float));
                auto it = seen.find(bits);
                if (it == seen.end()) {
                    const uint32_t id = static_cast<uint32_t>(seen.size());
                    if (id >= 65536u) { ok = false; break; }
                    wg2.resize(wg2.size() + row_floats);
                    std::memcpy(wg2.data() + static_cast<size_t>(id) * row_floats, bits.data(), row_floats * 4u);
                    it = seen.emplace(std::move(bits), id).first;
                }
                if ((p->wh.left[u] >> 1) >= 65536u) { ok = false; break; }
                hmeta3[u] = (p->wh.left[u] >> 1) | (it->second << 16);
            }
            if (ok) {
                while (wg2.size() & 3u) wg2.push_back(0.0f);             // (staged into LDS in 16-byte pieces)
                p->h_two_groups = g2_max; p->h_wg2_floats = static_cast<uint32_t>(wg2.size());
            }
        }
    }

    if ((rc = upload(p->wv.left, &p->d_v_left)) || (rc = upload(p->wv.count, &p->d_v_count)) ||
        (rc = upload(p->wv.offset, &p->d_v_off)) || (rc = upload(p->wv.w, &p->d_v_w)) ||
        (rc = upload(p->wh.left, &p->d_h_left)) || (rc = upload(p->wh.count, &p->d_h_count)) ||
        (rc = upload(p->wh.offset, &p->d_h_off)) || (rc = upload(p->wh.w, &p->d_h_w)) || (rc = upload(wu, &p->d_h_wu)) || (rc = upload(hmeta, &p->d_h_meta)))
        return rc;
    if (p->h_fast_groups && ((rc = upload(wg, &p->d_h_wg)) || (rc = upload(hmeta2, &p->d_h_meta2)))) return rc;
    if (p->h_two_groups && ((rc = upload(wg2, &p->d_h_wg2)) || (rc = upload(hmeta3, &p->d_h_meta3)))) return rc;

    p->slots = max_live_rows(p->wv);
    VSchedule probe;
    p->fused_possible = p->slots >= 1 && p->slots <= kMaxSlots && build_vschedule(p->wv, 1, 4, 5, &probe);
    for (int al = 0; al < 2 && p->fused_possible; ++al) {
        const int channels = al ? 4 : 3;
        ifhip_resample_plan::StripSet& ss = p->sets[al];
        const uint32_t max_lanes = static_cast<uint32_t>(fused_max_quads(p->slots, channels));      // in 4-pixel groups
        ss.ok = plan_strips(p->wh, max_lanes, fused_shape(p->slots, channels).px, channels, &ss.strips, &ss.max_quads);
        if (ss.ok && (rc = upload(ss.strips, &ss.d_strips))) return rc;
    }
    *plan = p.release();
    return IFHIP_OK;
}

void ifhip_resample_plan_destroy(ifhip_resample_plan* plan) { delete plan; }

int ifhip_resample_plan_kernel_kind(const ifhip_resample_plan* plan, int in_alpha_meaningful) {
    return (plan && plan->fused_possible && plan->sets[in_alpha_meaningful ? 1 : 0].ok) ? 0 : 1;
}

// What the fast horizontal pass of the fused kernel would run for this plan: groups of four source columns per output
// (0: not available) and groups of two (0: not available or not fewer taps); see ifhip_resample_plan_create.
int ifhip_resample_plan_horizontal_groups(const ifhip_resample_plan* plan, uint32_t* four_column_groups, uint32_t* two_column_groups) {
    if (!plan) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null plan");
    if (four_column_groups) *four_column_groups = plan->h_fast_groups;
    if (two_column_groups) *two_column_groups = plan->h_two_groups;
    return IFHIP_OK;
}

int ifhip_scale_and_render_batch_device(const ifhip_resample_plan* plan, const uint8_t* d_in, size_t in_image_bytes,
                                        uint32_t in_stride, int in_alpha_meaningful, uint32_t n_images,
                                        uint8_t* d_canvas, size_t canvas_image_bytes, uint32_t canvas_w,
                                        uint32_t canvas_h, uint32_t canvas_stride, uint32_t x, uint32_t y,
                                        int working_space, int compositing, uint32_t matte_bgra, float* d_f32_dump,
                                        int force_kernel, void* hip_stream) {
    return enqueue_batch(plan, d_in, in_image_bytes, in_stride, in_alpha_meaningful, n_images, d_canvas,
                         canvas_image_bytes, canvas_w, canvas_h, canvas_stride, x, y, working_space, compositing,
                         matte_bgra, d_f32_dump, force_kernel, static_cast<hipStream_t>(hip_stream));
}

int ifhip_time_scale_and_render_batch_device(const ifhip_resample_plan* plan, const uint8_t* d_in,
                                             size_t in_image_bytes, uint32_t in_stride, int in_alpha_meaningful,
                                             uint32_t n_images, uint8_t* d_canvas, size_t canvas_image_bytes,
                                             uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride, uint32_t x,
                                             uint32_t y, int working_space, int compositing, uint32_t matte_bgra,
                                             int force_kernel, void* hip_stream, int launches,
                                             float* avg_ms_per_launch) {
    if (launches < 1 || !avg_ms_per_launch) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: launches/avg pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    EventPair ev;
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, st));
    int rc = IFHIP_OK;
    for (int i = 0; i < launches && rc == IFHIP_OK; ++i)
        rc = enqueue_batch(plan, d_in, in_image_bytes, in_stride, in_alpha_meaningful, n_images, d_canvas,
                           canvas_image_bytes, canvas_w, canvas_h, canvas_stride, x, y, working_space, compositing,
                           matte_bgra, nullptr, force_kernel, st);
    hipError_t er = hipEventRecord(ev.e1, st);
    if (er == hipSuccess) er = hipEventSynchronize(ev.e1);
    float ms = 0.f;
    if (er == hipSuccess) er = hipEventElapsedTime(&ms, ev.e0, ev.e1);
    if (rc) return rc;
    if (er != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: event timing failed: %s", hipGetErrorString(er));
    *avg_ms_per_launch = ms / static_cast<float>(launches);
    return IFHIP_OK;
}

int ifhip_measure_copy_bandwidth(size_t bytes, int iters, double* bytes_per_second) {
    if (!bytes_per_second || iters < 1 || bytes == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: copy bandwidth probe");
    DeviceTables tb;
    int rc = device_tables(&tb);
    if (rc) return rc;
    DeviceBuffer a, b;
    HIP_TRY(a.alloc(bytes));
    HIP_TRY(b.alloc(bytes));
    HIP_TRY(hipMemset(a.p, 1, bytes));
    HIP_TRY(hipMemcpy(b.p, a.p, bytes, hipMemcpyDeviceToDevice));
    EventPair ev;
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, nullptr));
    for (int i = 0; i < iters; ++i) HIP_TRY(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, nullptr));
    HIP_TRY(hipEventRecord(ev.e1, nullptr));
    HIP_TRY(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *bytes_per_second = 2.0 * static_cast<double>(bytes) * iters / (static_cast<double>(ms) * 1e-3);
    return IFHIP_OK;
}

int ifhip_measure_read_bandwidth(size_t bytes, int iters, double* bytes_per_second) {
    if (!bytes_per_second || iters < 1 || bytes < (1u << 20)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: read bandwidth probe");
    DeviceTables tb;
    int rc = device_tables(&tb);
    if (rc) return rc;
    bytes &= ~static_cast<size_t>(4095);
    DeviceBuffer a, sink;
    HIP_TRY(a.alloc(bytes));
    HIP_TRY(sink.alloc(4096));
    HIP_TRY(hipMemset(a.p, 1, bytes));
    HIP_TRY(hipMemset(sink.p, 0, 4096));
    HIP_TRY(launch_read_probe(static_cast<const uint8_t*>(a.p), bytes, static_cast<uint32_t*>(sink.p), nullptr));   // warm-up
    EventPair ev;
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, nullptr));
    for (int i = 0; i < iters; ++i) HIP_TRY(launch_read_probe(static_cast<const uint8_t*>(a.p), bytes, static_cast<uint32_t*>(sink.p), nullptr));
    HIP_TRY(hipEventRecord(ev.e1, nullptr));
    HIP_TRY(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *bytes_per_second = static_cast<double>(bytes) * iters / (static_cast<double>(ms) * 1e-3);
    return IFHIP_OK;
}

int ifhip_measure_mixed_bandwidth(size_t read_bytes, uint32_t read_vectors_per_write, int iters, double* bytes_per_second) {
    if (!bytes_per_second || iters < 1 || read_bytes < (1u << 20) || read_vectors_per_write < 1u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: mixed bandwidth probe");
    DeviceTables tb;
    int rc = device_tables(&tb);
    if (rc) return rc;
    read_bytes &= ~static_cast<size_t>(4095);
    const size_t write_cap = read_bytes / read_vectors_per_write + (static_cast<size_t>(64) << 20);   // every workgroup's span keeps its own output span
    DeviceBuffer a, out, sink;
    HIP_TRY(a.alloc(read_bytes));
    HIP_TRY(out.alloc(std::max(write_cap, read_bytes)));
    HIP_TRY(sink.alloc(8192));
    HIP_TRY(hipMemset(a.p, 1, read_bytes));
    HIP_TRY(hipMemset(sink.p, 0, 8192));
    HIP_TRY(launch_mix_probe(static_cast<const uint8_t*>(a.p), static_cast<uint8_t*>(out.p), read_bytes, read_vectors_per_write, static_cast<uint32_t*>(sink.p), nullptr));   // warm-up
    EventPair ev;
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, nullptr));
    for (int i = 0; i < iters; ++i)
        HIP_TRY(launch_mix_probe(static_cast<const uint8_t*>(a.p), static_cast<uint8_t*>(out.p), read_bytes, read_vectors_per_write, static_cast<uint32_t*>(sink.p), nullptr));
    HIP_TRY(hipEventRecord(ev.e1, nullptr));
    HIP_TRY(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    uint32_t stores_per_lane = 0;
    HIP_TRY(hipMemcpy(&stores_per_lane, static_cast<const uint32_t*>(sink.p) + 1024, 4, hipMemcpyDeviceToHost));
    const double written = static_cast<double>(stores_per_lane) * 16.0 * 1024.0 * (256.0 * 8.0);
    *bytes_per_second = (static_cast<double>(read_bytes) + written) * iters / (static_cast<double>(ms) * 1e-3);
    return IFHIP_OK;
}

// ---- host-buffer drop-ins ---------------------------------------------------------------------------------
}  // extern "C"

namespace {
struct PlanKey {
    int device; uint32_t in_w, in_h, w, h; int filter; uint32_t sharpen_bits;
    bool operator<(const PlanKey& o) const {
        return std::tie(device, in_w, in_h, w, h, filter, sharpen_bits) < std::tie(o.device, o.in_w, o.in_h, o.w, o.h, o.filter, o.sharpen_bits);
    }
};
struct PlanEntry {
    std::shared_ptr<ifhip_resample_plan> plan;
    uint64_t last_use = 0;
};
std::mutex g_plan_mu;
std::map<PlanKey, PlanEntry> g_plan_cache;
uint64_t g_plan_clock = 0;
constexpr size_t kPlanCacheMax = 64;

int cached_plan(uint32_t in_w, uint32_t in_h, uint32_t w, uint32_t h, int filter, float sharpen,
                std::shared_ptr<ifhip_resample_plan>* out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device (hipGetDevice failed); this library has no CPU path");
    uint32_t bits;
    std::memcpy(&bits, &sharpen, 4);
    const PlanKey key{dev, in_w, in_h, w, h, filter, bits};
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_plan_cache.find(key);
        if (it != g_plan_cache.end()) { it->second.last_use = ++g_plan_clock; *out = it->second.plan; return IFHIP_OK; }
    }
    ifhip_resample_plan* raw = nullptr;
    const int rc = ifhip_resample_plan_create(&raw, in_w, in_h, w, h, filter, sharpen);
    if (rc) return rc;
    std::shared_ptr<ifhip_resample_plan> sp(raw);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_plan_cache.size() >= kPlanCacheMax) {             // least recently used shape goes (callers still holding it keep it alive)
        auto victim = g_plan_cache.begin();
        for (auto it = g_plan_cache.begin(); it != g_plan_cache.end(); ++it)
            if (it->second.last_use < victim->second.last_use) victim = it;
        g_plan_cache.erase(victim);
    }
    g_plan_cache[key] = PlanEntry{sp, ++g_plan_clock};
    *out = sp;
    return IFHIP_OK;
}

// ---- staging of the host-buffer drop-ins -------------------------------------------------------------
// imageflow runs "one Context per thread" (imageflow_abi/src/lib.rs:20-27), so every calling thread gets its own
// HIP stream, pinned host staging and HBM staging, all grow-only and kept between calls: no hipMalloc / hipFree /
// hipMemset on the call path, no null-stream serialisation between threads.  A thread's staging returns to a pool
// when the thread ends (no HIP call at thread exit) and is adopted by the next new thread.
struct HostStage {
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t *pin_in = nullptr, *pin_c = nullptr, *d_in = nullptr, *d_c = nullptr;
    size_t pin_in_cap = 0, pin_c_cap = 0, d_in_cap = 0, d_c_cap = 0;
};
std::mutex g_stage_mu;
std::vector<HostStage*> g_stage_pool;
struct StageLease {
    HostStage* s = nullptr;
    ~StageLease() {
        if (!s) return;
        std::lock_guard<std::mutex> lk(g_stage_mu);
        g_stage_pool.push_back(s);
    }
};
thread_local StageLease t_stage;

int grow_pinned(uint8_t** p, size_t* cap, size_t want) {
    if (*cap >= want) return IFHIP_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t sz = (want + (want >> 2) + 4095u) & ~static_cast<size_t>(4095);
    if (hipHostMalloc(reinterpret_cast<void**>(p), sz, hipHostMallocDefault) != hipSuccess)
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: %zu bytes of pinned host staging", sz);
    *cap = sz;
    return IFHIP_OK;
}
int grow_device(uint8_t** p, size_t* cap, size_t want) {
    if (*cap >= want) return IFHIP_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t sz = (want + (want >> 2) + 4095u) & ~static_cast<size_t>(4095);
    if (hipMalloc(reinterpret_cast<void**>(p), sz) != hipSuccess)
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: %zu bytes of HBM staging", sz);
    *cap = sz;
    return IFHIP_OK;
}

int host_stage(HostStage** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device (hipGetDevice failed); this library has no CPU path");
    HostStage*& s = t_stage.s;
    if (s && s->device != dev) {                       // the thread switched devices: hand the old staging back
        std::lock_guard<std::mutex> lk(g_stage_mu);
        g_stage_pool.push_back(s);
        s = nullptr;
    }
    if (!s) {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        for (size_t i = 0; i < g_stage_pool.size(); ++i)
            if (g_stage_pool[i]->device == dev) { s = g_stage_pool[i]; g_stage_pool.erase(g_stage_pool.begin() + static_cast<long>(i)); break; }
    }
    if (!s) {
        std::unique_ptr<HostStage> n(new HostStage);
        n->device = dev;
        HIP_TRY(hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking));
        s = n.release();
    }
    *out = s;
    return IFHIP_OK;
}

// Host -> HBM through the pinned buffer in chunks: the DMA of chunk k runs while the CPU copies chunk k+1.
hipError_t upload_chunked(HostStage* s, uint8_t* d_dst, uint8_t* pinned, const uint8_t* src, size_t bytes) {
    constexpr size_t kChunk = 4u << 20;
    for (size_t off = 0; off < bytes; off += kChunk) {
        const size_t n = std::min(kChunk, bytes - off);
        std::memcpy(pinned + off, src + off, n);
        const hipError_t e = hipMemcpyAsync(d_dst + off, pinned + off, n, hipMemcpyHostToDevice, s->stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

extern "C" {
int ifhip_scale_and_render(const uint8_t* in, uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful,
                           uint8_t* canvas, uint32_t canvas_w, uint32_t canvas_h, uint32_t canvas_stride,
                           int /*canvas_alpha_meaningful*/, uint32_t x, uint32_t y, uint32_t w, uint32_t h, int filter,
                           float sharpen_percent_goal, int working_space, int compositing, uint32_t matte_bgra) {
    int rc = validate_render(in_w, in_h, in_stride, canvas_w, canvas_h, canvas_stride, x, y, w, h, working_space, compositing);
    if (rc) return rc;
    if (!in || !canvas) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null bitmap pointer");
    // plans are cached per (device, shape, filter, sharpen): a server resizing many frames of the same size pays for the
    // weight tables and the vertical schedule once
    std::shared_ptr<ifhip_resample_plan> plan_ref;
    rc = cached_plan(in_w, in_h, w, h, filter, sharpen_percent_goal, &plan_ref);
    if (rc) return rc;
    ifhip_resample_plan* plan = plan_ref.get();
    HostStage* s = nullptr;
    if ((rc = host_stage(&s))) return rc;
    // stage: source rows as given; only the canvas rows the rect touches
    const size_t in_bytes = (static_cast<size_t>(in_h) * in_stride + 15u) & ~static_cast<size_t>(15);
    const size_t in_valid = static_cast<size_t>(in_h - 1) * in_stride + static_cast<size_t>(in_w) * 4u;
    const size_t c_rows_bytes = (static_cast<size_t>(h) * canvas_stride + 3u) & ~static_cast<size_t>(3);
    const size_t c_valid = static_cast<size_t>(h - 1) * canvas_stride + static_cast<size_t>(canvas_w) * 4u;
    if ((rc = grow_pinned(&s->pin_in, &s->pin_in_cap, in_valid)) || (rc = grow_pinned(&s->pin_c, &s->pin_c_cap, c_valid)) ||
        (rc = grow_device(&s->d_in, &s->d_in_cap, in_bytes + 64)) || (rc = grow_device(&s->d_c, &s->d_c_cap, c_rows_bytes + 64)))
        return rc;
    uint8_t* crow0 = canvas + static_cast<size_t>(y) * canvas_stride;
    hipError_t e = upload_chunked(s, s->d_in, s->pin_in, in, in_valid);
    // the kernels read whole 16-byte groups up to the row stride: the tail of the last row gets defined bytes
    if (e == hipSuccess) e = hipMemsetAsync(s->d_in + in_valid, 0, in_bytes + 64 - in_valid, s->stream);
    // the canvas rows travel to the device only when the call can leave some of their bytes untouched or reads them
    const bool canvas_needed = compositing == IFHIP_BLEND_WITH_SELF || x != 0 || w != canvas_w;
    if (e == hipSuccess && canvas_needed) e = upload_chunked(s, s->d_c, s->pin_c, crow0, c_valid);
    if (e == hipSuccess) {
        rc = enqueue_batch(plan, s->d_in, in_bytes, in_stride, in_alpha_meaningful, 1, s->d_c, c_rows_bytes,
                           canvas_w, h, canvas_stride, x, 0, working_space, compositing, matte_bgra, nullptr, -1, s->stream);
        if (rc == IFHIP_OK) {
            e = hipMemcpyAsync(s->pin_c, s->d_c, c_valid, hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = static_cast<hipError_t>(ifhip::wait_stream(s->stream));
            if (e == hipSuccess) {
                if (canvas_needed) std::memcpy(crow0, s->pin_c, c_valid);
                else                                      // whole rows were produced: leave the caller's row padding alone
                    for (uint32_t j = 0; j < h; ++j)
                        std::memcpy(crow0 + static_cast<size_t>(j) * canvas_stride, s->pin_c + static_cast<size_t>(j) * canvas_stride,
                                    static_cast<size_t>(canvas_w) * 4u);
            }
        } else (void)static_cast<hipError_t>(ifhip::wait_stream(s->stream));
    } else (void)static_cast<hipError_t>(ifhip::wait_stream(s->stream));
    if (rc) return rc;
    if (e != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: staging failed: %s", hipGetErrorString(e));
    return IFHIP_OK;
}

int ifhip_apply_matte_batch_device(uint8_t* d_bgra, size_t image_bytes, uint32_t n_images, uint32_t w, uint32_t h,
                                   uint32_t stride, int alpha_meaningful, uint32_t matte_bgra, void* hip_stream) {
    if (!alpha_meaningful) return IFHIP_OK;                       // blend.rs:11-13
    if (w == 0 || h == 0 || n_images == 0) return IFHIP_OK;
    if (!d_bgra) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null bitmap pointer");
    if (static_cast<uint64_t>(w) * 4u > stride || (stride & 3u) || (image_bytes & 3u) || (reinterpret_cast<uintptr_t>(d_bgra) & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: BGRA rows must be 4-byte aligned and stride >= 4*w");
    DeviceTables tb;
    int rc = device_tables(&tb);
    if (rc) return rc;
    const ColorTables& t = color_tables();
    HIP_TRY(launch_apply_matte(d_bgra, image_bytes, n_images, w, h, stride, matte_bgra, t.s2l[matte_bgra & 255u],
                               t.s2l[(matte_bgra >> 8) & 255u], t.s2l[(matte_bgra >> 16) & 255u],
                               static_cast<float>(matte_bgra >> 24) * (1.0f / 255.0f), tb.s2l, tb.l2s,
                               static_cast<hipStream_t>(hip_stream)));
    return IFHIP_OK;
}

int ifhip_apply_matte(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int alpha_meaningful, uint32_t matte_bgra) {
    if (!alpha_meaningful) return IFHIP_OK;
    if (w == 0 || h == 0) return IFHIP_OK;
    if (!bgra) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null bitmap pointer");
    if (static_cast<uint64_t>(w) * 4u > stride || (stride & 3u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: BGRA rows must be 4-byte aligned and stride >= 4*w");
    const size_t valid = static_cast<size_t>(h - 1) * stride + static_cast<size_t>(w) * 4u;
    const size_t bytes = (static_cast<size_t>(h) * stride + 15u) & ~static_cast<size_t>(15);
    HostStage* s = nullptr;
    int rc = host_stage(&s);
    if (rc) return rc;
    if ((rc = grow_pinned(&s->pin_c, &s->pin_c_cap, valid)) || (rc = grow_device(&s->d_c, &s->d_c_cap, bytes + 64))) return rc;
    hipError_t e = upload_chunked(s, s->d_c, s->pin_c, bgra, valid);
    if (e == hipSuccess) {
        rc = ifhip_apply_matte_batch_device(s->d_c, bytes, 1, w, h, stride, 1, matte_bgra, s->stream);
        if (rc == IFHIP_OK) {
            e = hipMemcpyAsync(s->pin_c, s->d_c, valid, hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = static_cast<hipError_t>(ifhip::wait_stream(s->stream));
            if (e == hipSuccess) std::memcpy(bgra, s->pin_c, valid);
        } else (void)static_cast<hipError_t>(ifhip::wait_stream(s->stream));
    } else (void)static_cast<hipError_t>(ifhip::wait_stream(s->stream));
    if (rc) return rc;
    if (e != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: staging failed: %s", hipGetErrorString(e));
    return IFHIP_OK;
}

}  // extern "C"
