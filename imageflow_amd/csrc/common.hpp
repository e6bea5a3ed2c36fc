// common.hpp -- shared declarations for libimageflow_hip.so (host side).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/imageflow_hip.h"

namespace ifhip {

// thread-local last error (FlowError.message analogue)
int fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error();

// ---- memory and stream of one job's objects (devmem.cpp) ---------------------------------------------------------------
// Size-class caches in front of hipMalloc / hipHostMalloc: plans, stages, entropy handles and frames are created and
// destroyed once per JOB, and neither call may cost a driver round trip (or hipFree's wait for the whole device) there.
// Return values are hipError_t as int (this header stays free of hip_runtime.h).  cached_free waits for the device like
// hipFree unless the caller holds a QuiescedScope: "the stream my work ran on has been synchronised".
int cached_malloc(void** out, size_t bytes);
int cached_free(void* p);
int cached_malloc_for_stream(void** out, size_t bytes, void* stream, bool for_stream);   // a block for work on `stream` only
int cached_free_after(void* p, void* stream);           // back to the cache once the work queued on `stream` so far has run; no host wait
int cached_host_malloc(void** out, size_t bytes);       // pinned, portable
int cached_host_free(void* p);
void quiesced_enter();
void quiesced_leave();
struct QuiescedScope { QuiescedScope() { quiesced_enter(); } ~QuiescedScope() { quiesced_leave(); } };
// The stream a thread's create paths upload / clear on (ifhip_set_thread_stream; default: the null stream), and copies on it
// that are complete on return.
void* thread_stream();
// Every wait of a job for its stream goes through here: load-aware -- the first few waiters (a quarter of the CPUs the process
// may use) spin in the runtime, the rest query the stream and sleep in between (devmem.cpp).
int wait_stream(void* hip_stream);
int copy_to_device(void* dst, const void* src, size_t bytes);
int stage_to_device(void* dst, const void* src, size_t bytes, void** pin_out);   // queued, NOT waited for; *pin_out: cached_host_free after the stream's next wait
int copy_to_host(void* dst, const void* src, size_t bytes);
int zero_device(void* dst, size_t bytes);
int require_gfx950(int* device_out);                  // IFHIP_OK and the current device, or GpuUnavailable (asked of the driver once)
#define DEV_MALLOC(pp, n) static_cast<hipError_t>(::ifhip::cached_malloc(reinterpret_cast<void**>(pp), (n)))
#define DEV_FREE(p) static_cast<hipError_t>(::ifhip::cached_free(p))

// Development switches (tests and tools/ only).  The library never reads the environment: a switch exists only after
// ifhip_debug_set(key, value) (include/imageflow_hip.h); unset -> nullptr.  One relaxed atomic load when none is set.
const char* debug_switch(const char* key);
inline bool debug_on(const char* key) { return debug_switch(key) != nullptr; }

// ---------------------------------------------------------------------------------------------------
// Interpolation kernels + per-axis contribution tables  (graphics/weights.rs)
// ---------------------------------------------------------------------------------------------------
enum class KernelShape : uint8_t { FlexCubic, CubicFast, Sinc, Box, Triangle, SincWindowed, Jinc, Ginseng };

struct FilterSpec {            // InterpolationDetails (weights.rs:107-124) minus the fn pointer
    double window = 2.0;
    double blur = 1.0;
    double p1 = 0, p2 = 1, p3 = 1, q1 = 0, q2 = 1, q3 = 1, q4 = 1;
    KernelShape shape = KernelShape::Box;
    int lobe_mode = IFHIP_LOBE_NATURAL;
    float lobe_value = 0.f;

    double eval(double x) const;                 // (self.filter)(self, x)
    double natural_negative_ratio() const;       // calculate_percent_negative_weight, weights.rs:333-350
};

bool filter_spec_for(int filter, FilterSpec* out);     // InterpolationDetails::create, weights.rs:176-331

struct AxisWeights {           // PixelRowWeights (weights.rs:521-571) in structure-of-arrays form
    uint32_t n_out = 0, n_in = 0;
    std::vector<uint32_t> left;     // left_pixel
    std::vector<uint32_t> count;    // right_pixel - left_pixel + 1
    std::vector<uint32_t> offset;   // left_weight
    std::vector<float> w;
    uint32_t max_taps = 0;
};

// populate_weights, weights.rs:681-788.  Returns IFHIP_OK or an error status (message set).
int build_axis_weights(const FilterSpec& spec, uint32_t out_size, uint32_t in_size, AxisWeights* out);

// ---------------------------------------------------------------------------------------------------
// Colour tables  (graphics/color.rs, graphics/lut.rs)
// ---------------------------------------------------------------------------------------------------
struct ColorTables {
    float s2l[256];        // ColorContext(LinearRGB).byte_to_float
    float s2f[256];        // ColorContext(StandardRGB).byte_to_float
    uint8_t l2s[16384];    // LINEAR_TO_SRGB_LUT
    uint16_t l2s_thr[256]; // thr[k] = smallest i with l2s[i] >= k+1 (65535 if none): l2s[i] == #{k : thr[k] <= i}
};
const ColorTables& color_tables();

// ---------------------------------------------------------------------------------------------------
// Vertical schedule for the fused kernel (our own construct; DESIGN.md "vertical schedule")
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxSlots = 8;
struct alignas(64) VStep {
    int32_t y;            // source row to load and accumulate, or -1
    uint32_t active;      // bit s set: ring slot s accumulates row y with weight w[s]
    int32_t flush_slot;   // ring slot completed by this step (-1: none)
    int32_t out_row;      // output row index held by flush_slot
    float w[kMaxSlots];
    int32_t y_ahead;      // source row of step i+ahead of the same band (-1: none): the load issued at step i
    int32_t pad[3];
};
static_assert(sizeof(VStep) == 64, "VStep must be one 64-byte scalar-load line");

struct VSchedule {
    int slots = 0;                       // ring size K = max simultaneously live output rows
    std::vector<VStep> steps;            // concatenated over bands
    std::vector<uint32_t> band_begin;    // n_bands + 1 offsets into steps
};
// Split out rows [0, n_out) into n_bands contiguous bands and emit the step list of each.
// Returns false if more than kMaxSlots rows are live at once (caller uses the generic kernels).
// `group`: bands are padded to a multiple of `group` steps (the kernel's unroll); `ahead`: steps[i].y_ahead = row of step
// i + ahead (the load the kernel issues while working on step i).
bool build_vschedule(const AxisWeights& wv, int n_bands, int group, int ahead, VSchedule* out);

// internal status of the planar-source resample call: this shape / these pointers cannot run on the fused kernel
constexpr int kNotFusable = -1000;
// (api.cpp; hipStream_t passes as void* so that this header stays free of hip_runtime.h)
int resample_from_ycc_planes_v(const ifhip_resample_plan* plan, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr,
                               size_t plane_bytes, uint32_t pitch, uint32_t n_images, uint8_t* d_canvas, size_t canvas_image_bytes,
                               uint32_t cw, uint32_t ch, uint32_t c_stride, uint32_t x, uint32_t y, int working_space, int compositing,
                               uint32_t matte, void* hip_stream, bool probe = false);   // probe: decide only, launch nothing
void resample_plan_shape(const ifhip_resample_plan* plan, uint32_t* in_w, uint32_t* in_h, uint32_t* out_w, uint32_t* out_h);
int max_live_rows(const AxisWeights& wv);

}  // namespace ifhip
