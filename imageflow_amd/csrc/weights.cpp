// weights.cpp -- host-side filter catalogue and contribution tables.
//
// Follows imageflow_core/src/graphics/weights.rs: Filter discriminants :43-78, InterpolationDetails::create
// :176-331, calculate_percent_negative_weight :333-350, kernel functions :352-458, bessj1 :460-492,
// populate_weights :681-788.  All evaluation in f64, storage in f32, exactly where the reference narrows.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>

#include "common.hpp"

namespace ifhip {

static thread_local char g_err[512] = "";

int fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}
const char* last_error() { return g_err; }

// development switches: a small registry behind ifhip_debug_set; values live until the process ends (a changed value
// leaves the old string in place, so a pointer handed out earlier stays readable)
namespace {
std::atomic<int> g_switches_set{0};
std::mutex g_switch_mu;
std::map<std::string, const std::string*>& switches() { static std::map<std::string, const std::string*> m; return m; }
}  // namespace
const char* debug_switch(const char* key) {
    if (g_switches_set.load(std::memory_order_relaxed) == 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_switch_mu);
    const auto it = switches().find(key);
    return it == switches().end() || !it->second ? nullptr : it->second->c_str();
}

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kRobidouxB = 0.3782157550939987, kRobidouxC = 0.3108921224530007;
constexpr double kSharpB = 0.2620145123990142, kSharpC = 0.3689927438004929;
constexpr double kBlurL2Sharp = 0.9549963639785485, kBlurL3Sharp = 0.9812505644269356;

FilterSpec windowed(KernelShape s, double window, double blur) {
    FilterSpec f;                 // p/q keep the Default::default() values of weights.rs:126-142
    f.shape = s; f.window = window; f.blur = blur;
    return f;
}
FilterSpec bc(double window, double blur, double b, double c) {   // InterpolationDetails::bicubic :159-174
    FilterSpec f;
    const double b2 = b + b;
    f.shape = KernelShape::FlexCubic; f.window = window; f.blur = blur;
    f.p1 = 1.0 - (1.0 / 3.0) * b;
    f.p2 = -3.0 + b2 + c;
    f.p3 = 2.0 - 1.5 * b - c;
    f.q1 = (4.0 / 3.0) * b + 4.0 * c;
    f.q2 = -8.0 * c - b2;
    f.q3 = b + 5.0 * c;
    f.q4 = (-1.0 / 6.0) * b - c;
    return f;
}

double bessel_j1(double x) {     // weights.rs:460-492 (Numerical-Recipes rational fits)
    const double ax = std::fabs(x);
    double r;
    if (ax < 8.0) {
        const double y = x * x;
        const double num = x * (72362614232.0 + y * (-7895059235.0 + y * (242396853.1 + y * (-2972611.439
                           + y * (15704.48260 + y * (-30.16036606))))));
        const double den = 144725228442.0 + y * (2300535178.0 + y * (18583304.74 + y * (99447.43394
                           + y * (376.9991397 + y * 1.0))));
        r = num / den;
    } else {
        const double z = 8.0 / ax, y = z * z, xx = ax - 2.356194491;
        const double a1 = 1.0 + y * (0.183105e-2 + y * (-0.3516396496e-4 + y * (0.2457520174e-5 + y * (-0.240337019e-6))));
        const double a2 = 0.04687499995 + y * (-0.2002690873e-3 + y * (0.8449199096e-5 + y * (-0.88228987e-6
                          + y * 0.105787412e-6)));
        r = std::sqrt(0.63661977236758134307553505349005744 / ax) * (std::cos(xx) * a1 - z * std::sin(xx) * a2);
    }
    return x < 0.0 ? -r : r;
}

int32_t to_i32_saturating(double v) {      // Rust `as i32`
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return static_cast<int32_t>(v);
}

}  // namespace

double FilterSpec::eval(double x) const {
    switch (shape) {
    case KernelShape::FlexCubic: {
        const double t = std::fabs(x) / blur;
        if (t < 1.0) return p1 + t * (t * (p2 + t * p3));
        if (t < 2.0) return q1 + t * (q2 + t * (q3 + t * q4));
        return 0.0;
    }
    case KernelShape::CubicFast: {
        const double a = std::fabs(x) / blur, a2 = a * a;
        if (a < 1.0) return 1.0 - 2.0 * a2 + a2 * a;
        if (a < 2.0) return 4.0 - 8.0 * a + 5.0 * a2 - a2 * a;
        return 0.0;
    }
    case KernelShape::Sinc: {
        const double a = std::fabs(x) / blur;
        if (a == 0.0) return 1.0;
        if (a > window) return 0.0;
        const double ap = a * kPi;
        return std::sin(ap) / ap;
    }
    case KernelShape::Box: {
        const double t = x / blur;
        return (t >= -window && t < window) ? 1.0 : 0.0;
    }
    case KernelShape::Triangle: {
        const double t = std::fabs(x) / blur;
        return t < 1.0 ? 1.0 - t : 0.0;
    }
    case KernelShape::SincWindowed: {
        const double t = x / blur, a = std::fabs(t);
        if (a == 0.0) return 1.0;
        if (a > window) return 0.0;
        return window * std::sin(kPi * t / window) * std::sin(t * kPi) / (kPi * kPi * t * t);
    }
    case KernelShape::Jinc: {
        const double t = std::fabs(x) / blur;
        if (t == 0.0) return 0.5 * kPi;
        return bessel_j1(kPi * t) / t;
    }
    case KernelShape::Ginseng: {
        const double a = std::fabs(x) / blur, tp = a * kPi;
        if (a == 0.0) return 1.0;
        if (a > 3.0) return 0.0;
        const double ji = 1.2196698912665046 * tp / window;
        return (bessel_j1(ji) / (ji * 0.5)) * std::sin(tp) / tp;
    }
    }
    return 0.0;
}

double FilterSpec::natural_negative_ratio() const {
    constexpr int samples = 50;
    const double step = window / samples;
    double prev = eval(-step), pos = 0.0, neg = 0.0;
    for (int i = 0; i < samples + 3; ++i) {
        const double h = eval(i * step);
        const double area = (h + prev) / 2.0 * step;
        prev = h;
        if (area > 0.0) pos += area; else neg -= area;
    }
    return neg / pos;
}

bool filter_spec_for(int filter, FilterSpec* out) {
    using K = KernelShape;
    switch (filter) {
    case IFHIP_FILTER_ROBIDOUX_FAST:        *out = bc(1.05, 1.0, kRobidouxB, kRobidouxC); return true;
    case IFHIP_FILTER_ROBIDOUX:             *out = bc(2.0, 1.0, kRobidouxB, kRobidouxC); return true;
    case IFHIP_FILTER_ROBIDOUX_SHARP:       *out = bc(2.0, 1.0, kSharpB, kSharpC); return true;
    case IFHIP_FILTER_GINSENG:              *out = windowed(K::Ginseng, 3.0, 1.0); return true;
    case IFHIP_FILTER_GINSENG_SHARP:        *out = windowed(K::Ginseng, 3.0, kBlurL3Sharp); return true;
    case IFHIP_FILTER_LANCZOS:              *out = windowed(K::SincWindowed, 3.0, 1.0); return true;
    case IFHIP_FILTER_LANCZOS_SHARP:        *out = windowed(K::SincWindowed, 3.0, kBlurL3Sharp); return true;
    case IFHIP_FILTER_LANCZOS2:             *out = windowed(K::SincWindowed, 2.0, 1.0); return true;
    case IFHIP_FILTER_LANCZOS2_SHARP:       *out = windowed(K::SincWindowed, 2.0, kBlurL2Sharp); return true;
    case IFHIP_FILTER_CUBIC_FAST:           *out = windowed(K::CubicFast, 2.0, 1.0); return true;
    case IFHIP_FILTER_CUBIC:                *out = bc(2.0, 1.0, 0.0, 1.0); return true;
    case IFHIP_FILTER_CUBIC_SHARP:          *out = bc(2.0, kBlurL2Sharp, 0.0, 1.0); return true;
    case IFHIP_FILTER_CATMULL_ROM:          *out = bc(2.0, 1.0, 0.0, 0.5); return true;
    case IFHIP_FILTER_MITCHELL:             *out = bc(2.0, 1.0, 1.0 / 3.0, 1.0 / 3.0); return true;
    case IFHIP_FILTER_CUBIC_B_SPLINE:       *out = bc(2.0, 1.0, 1.0, 0.0); return true;
    case IFHIP_FILTER_HERMITE:              *out = bc(1.0, 1.0, 0.0, 0.0); return true;
    case IFHIP_FILTER_JINC:                 *out = windowed(K::Jinc, 6.0, 1.0); return true;
    case IFHIP_FILTER_RAW_LANCZOS3:         *out = windowed(K::Sinc, 3.0, 1.0); return true;
    case IFHIP_FILTER_RAW_LANCZOS3_SHARP:   *out = windowed(K::Sinc, 3.0, kBlurL3Sharp); return true;
    case IFHIP_FILTER_RAW_LANCZOS2:         *out = windowed(K::Sinc, 2.0, 1.0); return true;
    case IFHIP_FILTER_RAW_LANCZOS2_SHARP:   *out = windowed(K::Sinc, 2.0, kBlurL2Sharp); return true;
    case IFHIP_FILTER_TRIANGLE:
    case IFHIP_FILTER_LINEAR:               *out = windowed(K::Triangle, 1.0, 1.0); return true;
    case IFHIP_FILTER_BOX:                  *out = windowed(K::Box, 0.5, 1.0); return true;
    case IFHIP_FILTER_CATMULL_ROM_FAST:     *out = bc(1.0, 1.0, 0.0, 0.5); return true;
    case IFHIP_FILTER_CATMULL_ROM_FAST_SHARP: *out = bc(1.0, 13.0 / 16.0, 0.0, 0.5); return true;
    case IFHIP_FILTER_FASTEST:              *out = bc(0.74, 0.74, kRobidouxB, kRobidouxC); return true;
    case IFHIP_FILTER_MITCHELL_FAST:        *out = bc(1.0, 1.0, 1.0 / 3.0, 1.0 / 3.0); return true;
    case IFHIP_FILTER_N_CUBIC:              *out = bc(2.5, 1.0 / 1.1685777620836933, kRobidouxB, kRobidouxC); return true;
    case IFHIP_FILTER_N_CUBIC_SHARP:        *out = bc(2.5, 1.0 / 1.105822933719019, kSharpB, kSharpC); return true;
    case IFHIP_FILTER_LEGACY_IDCT:          *out = bc(2.0, 1. / 1.1685777620836932, kRobidouxB, kRobidouxC); return true;
    default: return false;
    }
}

int build_axis_weights(const FilterSpec& spec, uint32_t out_size, uint32_t in_size, AxisWeights* out) {
    if (out_size == 0 || in_size == 0)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: weight table with a zero line size (%u -> %u)", in_size, out_size);
    const double natural = spec.natural_negative_ratio();
    double desired = natural;                                   // LobeRatio::resolve, weights.rs:33-39
    if (spec.lobe_mode == IFHIP_LOBE_EXACT) {
        desired = std::fmin(std::fmax(static_cast<double>(spec.lobe_value), 0.0), 1.0);
    } else if (spec.lobe_mode == IFHIP_LOBE_SHARPEN_PERCENT) {
        desired = std::fmin(1.0, std::fmax(natural, static_cast<double>(spec.lobe_value) / 100.0));
    }
    const double scale = static_cast<double>(out_size) / static_cast<double>(in_size);
    const double down = std::fmin(1.0, scale);
    const double half_window = (spec.window + 0.5) / down;
    const uint32_t cap = static_cast<uint32_t>(to_i32_saturating(std::ceil(2.0 * (half_window - 0.00001))) + 1);
    const double reach = spec.window / down;
    const bool rebalance_lobes = std::fabs(desired - natural) > 1e-10;

    AxisWeights r;
    r.n_out = out_size; r.n_in = in_size;
    r.left.resize(out_size); r.count.resize(out_size); r.offset.resize(out_size);
    r.w.reserve(static_cast<size_t>(out_size) * cap);
    std::vector<float> taps(cap);

    for (uint32_t u = 0; u < out_size; ++u) {
        const double center = (u + 0.5) / scale - 0.5;
        const int32_t lo_edge = to_i32_saturating(std::ceil(center - reach - 0.0001));
        const int32_t hi_edge = to_i32_saturating(std::floor(center + reach + 0.0001));
        const uint32_t first = static_cast<uint32_t>(lo_edge > 0 ? lo_edge : 0);
        const int32_t last_allowed = static_cast<int32_t>(in_size) - 1;
        const uint32_t last = static_cast<uint32_t>(hi_edge < last_allowed ? hi_edge : last_allowed);
        const uint32_t n = last - first + 1u;
        if (n > cap)
            return fail(IFHIP_INVALID_STATE, "InvalidState: SourcePixelCountTooLarge (%u > %u) at output %u", n, cap, u);

        double sum = 0.0, sum_neg = 0.0, sum_pos = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            double v = spec.eval(down * (static_cast<double>(first + i) - center));
            if (std::fabs(v) <= 2e-8) v = 0.0;
            taps[i] = static_cast<float>(v);
            sum += v;
            sum_neg += std::fmin(v, 0.0);
            sum_pos += std::fmax(v, 0.0);
        }
        float scale_neg = static_cast<float>(1.0 / sum);
        float scale_pos = scale_neg;
        if (sum <= 0.0 || rebalance_lobes) {
            if (sum_neg < 0.0) {
                if (desired < 1.0) {
                    const double want_pos = 1.0 / (1.0 - desired);
                    const double want_neg = desired * -want_pos;
                    scale_pos = static_cast<float>(want_pos / sum_pos);
                    scale_neg = static_cast<float>(want_neg / sum_neg);
                }
            } else if (sum == 0.0) {
                return fail(IFHIP_INVALID_STATE, "InvalidState: TotalWeightZero at output %u (%u -> %u)", u, in_size, out_size);
            }
        }
        for (uint32_t i = 0; i < n; ++i) taps[i] *= (taps[i] < 0.f) ? scale_neg : scale_pos;

        uint32_t b = 0, e = n;                    // drop exact-zero ends (weights.rs:771-782)
        while (e > b && taps[e - 1] == 0.f) --e;
        while (b < e && taps[b] == 0.f) ++b;
        if (e == b)
            return fail(IFHIP_INVALID_STATE, "InvalidState: NoPixelInputs at output %u (%u -> %u)", u, in_size, out_size);
        r.left[u] = first + b;
        r.count[u] = e - b;
        r.offset[u] = static_cast<uint32_t>(r.w.size());
        r.w.insert(r.w.end(), taps.begin() + b, taps.begin() + e);
        if (e - b > r.max_taps) r.max_taps = e - b;
    }
    *out = std::move(r);
    return IFHIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// Vertical schedule
// ---------------------------------------------------------------------------------------------------
int max_live_rows(const AxisWeights& wv) {
    // windows are monotone in j, so the live set at source row y is a contiguous range of j
    int best = 0;
    uint32_t lo = 0;
    for (uint32_t j = 0; j < wv.n_out; ++j) {
        // rows lo..j are live at y = left[j] if their right edge >= left[j]
        while (lo < j && wv.left[lo] + wv.count[lo] - 1 < wv.left[j]) ++lo;
        const int live = static_cast<int>(j - lo + 1);
        if (live > best) best = live;
    }
    return best;
}

bool build_vschedule(const AxisWeights& wv, int n_bands, int group, int ahead, VSchedule* out) {
    if (group < 1 || ahead < 1) return false;
    const int K = max_live_rows(wv);
    if (K > kMaxSlots) return false;
    // monotonicity is what makes `j % K` a valid ring assignment; verify instead of assuming
    for (uint32_t j = 1; j < wv.n_out; ++j) {
        if (wv.left[j] < wv.left[j - 1]) return false;
        if (wv.left[j] + wv.count[j] < wv.left[j - 1] + wv.count[j - 1]) return false;
    }
    if (n_bands < 1) n_bands = 1;
    if (static_cast<uint32_t>(n_bands) > wv.n_out) n_bands = static_cast<int>(wv.n_out);
    VSchedule s;
    s.slots = K;
    s.band_begin.push_back(0);
    for (int b = 0; b < n_bands; ++b) {
        const uint32_t j0 = static_cast<uint32_t>(static_cast<uint64_t>(wv.n_out) * b / n_bands);
        const uint32_t j1 = static_cast<uint32_t>(static_cast<uint64_t>(wv.n_out) * (b + 1) / n_bands);
        const uint32_t y0 = wv.left[j0];
        const uint32_t y1 = wv.left[j1 - 1] + wv.count[j1 - 1] - 1;
        uint32_t jlo = j0;                       // first row of the band not yet flushed
        for (uint32_t y = y0; y <= y1; ++y) {
            VStep st;
            std::memset(&st, 0, sizeof st);
            st.y = static_cast<int32_t>(y);
            st.flush_slot = -1; st.out_row = -1;
            bool any = false;
            std::vector<uint32_t> done;
            for (uint32_t j = jlo; j < j1 && wv.left[j] <= y; ++j) {
                const uint32_t last = wv.left[j] + wv.count[j] - 1;
                if (y > last) continue;
                const int slot = static_cast<int>(j % K);
                st.active |= 1u << slot;
                st.w[slot] = wv.w[wv.offset[j] + (y - wv.left[j])];
                any = true;
                if (y == last) done.push_back(j);
            }
            if (!any && done.empty()) continue;          // row feeds nothing in this band: never loaded
            if (!done.empty()) { st.flush_slot = static_cast<int32_t>(done[0] % K); st.out_row = static_cast<int32_t>(done[0]); }
            s.steps.push_back(st);
            for (size_t i = 1; i < done.size(); ++i) {    // extra completions of the same source row: flush-only steps
                VStep f;
                std::memset(&f, 0, sizeof f);
                f.y = -1; f.flush_slot = static_cast<int32_t>(done[i] % K); f.out_row = static_cast<int32_t>(done[i]);
                s.steps.push_back(f);
            }
            while (jlo < j1 && wv.left[jlo] + wv.count[jlo] - 1 <= y) ++jlo;
        }
        while ((s.steps.size() - s.band_begin.back()) % static_cast<size_t>(group) != 0) {   // kernel consumes steps in groups
            VStep nop;
            std::memset(&nop, 0, sizeof nop);
            nop.y = -1; nop.flush_slot = -1; nop.out_row = -1; nop.y_ahead = -1;
            s.steps.push_back(nop);
        }
        const size_t begin = s.band_begin.back(), end = s.steps.size();
        for (size_t i = begin; i < end; ++i) {
            const size_t a = i + static_cast<size_t>(ahead);
            s.steps[i].y_ahead = a < end ? s.steps[a].y : -1;
        }
        s.band_begin.push_back(static_cast<uint32_t>(s.steps.size()));
    }
    *out = std::move(s);
    return true;
}

}  // namespace ifhip

extern "C" int ifhip_debug_set(const char* key, const char* value) {
    if (!key || !*key) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null switch name");
    std::lock_guard<std::mutex> lk(ifhip::g_switch_mu);
    ifhip::switches()[key] = value ? new std::string(value) : nullptr;
    ifhip::g_switches_set.store(1, std::memory_order_relaxed);
    return IFHIP_OK;
}
