// abi_shim.cpp -- the OUTER drop-in boundary: a subset of libimageflow's C ABI v3.2 (imageflow_abi/src/lib.rs:389-1496,
// header bindings/headers/imageflow_default.h) over the gfx950 kernels, declared in include/imageflow_abi_subset.h.
//
// Deliberately thin: a context with an io table, a sticky error and a response list; a small
// JSON reader; and a straight-line interpreter for the `v1/build` / `v1/execute` job shapes that reach the pixel hot
// path -- decode(baseline JPEG) -> [orientation / crop / canvas primitives] -> resample_2d | constrain | command_string
// -> encode -- as `steps` or as a `graph` with `input` edges (one producer per node, any number of consumers).  It is
// NOT imageflow's router or graph engine: nodes outside that list answer ActionNotSupported, and everything a job
// computes is computed by the ifhip_* entry points of this library on frames that stay in HBM.
//
// Two labelled EXTENSIONS, because the reference's JSON API has no raw-pixel I/O:
//   * decode accepts, besides baseline JPEG, the container "IFBGRA1\0" + u32le w, h, stride, alpha_meaningful + rows;
//   * encode writes a real JPEG for the libjpeg_turbo preset -- baseline, optimised tables, progressive (device pixel stage + host Huffman coder, jpeg_write.cpp)
//     and that container for every other preset (preferred_extension "ifbgra", mime "application/x-imageflow-bgra"):
//     PNG deflate / GIF / WebP coders are out of scope, the caller's encoder takes the frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/imageflow_abi_subset.h"
#include "../../include/imageflow_hip.h"
#include "common.hpp"           // the library's per-job memory cache and thread stream (devmem.cpp)
#include "layout.hpp"           // imageflow_riapi's constraint layout (constrain / watermark)

namespace {

// ---- errors: ErrorCategory (imageflow_core/src/errors.rs:779-838) with its exit / HTTP maps (:849-902) -----------
enum Cat { kOk = 0, kOutOfMemory = 1, kArgumentInvalid = 2, kInvalidJson = 3, kImageMalformed = 4, kImageTypeNotSupported = 5,
           kNodeArgumentInvalid = 6, kGraphInvalid = 7, kActionNotSupported = 8, kIoError = 16, kInternalError = 18,
           kOperationCancelled = 21 };
int http_code(int c) {
    switch (c) {
    case kOk: return 200;
    case kArgumentInvalid: case kGraphInvalid: case kNodeArgumentInvalid: case kActionNotSupported: case kInvalidJson:
    case kImageMalformed: case kImageTypeNotSupported: return 400;
    case kOutOfMemory: return 503;
    case kOperationCancelled: return 499;
    default: return 500;
    }
}
int exit_code(int c) {
    switch (c) {
    case kOk: return 0;
    case kArgumentInvalid: case kGraphInvalid: case kActionNotSupported: case kNodeArgumentInvalid: return 64;
    case kInvalidJson: case kImageMalformed: case kImageTypeNotSupported: return 65;
    case kOutOfMemory: return 71;
    case kIoError: return 74;
    case kOperationCancelled: return 130;
    default: return 70;
    }
}
struct FlowErr {
    int cat;
    std::string msg;
};
[[noreturn]] void raise(int cat, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void raise(int cat, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw FlowErr{cat, buf};
}
void check(int rc) {                                  // ifhip_status -> ErrorCategory
    if (rc == IFHIP_OK) return;
    const char* m = ifhip_last_error_message();
    const std::string msg = m ? m : "";
    int cat = kInternalError;
    if (rc == IFHIP_INVALID_ARGUMENT) cat = msg.rfind("ImageMalformed", 0) == 0 ? kImageMalformed : kArgumentInvalid;
    else if (rc == IFHIP_METHOD_NOT_IMPLEMENTED) cat = kActionNotSupported;
    else if (rc == IFHIP_ALLOCATION_FAILED) cat = kOutOfMemory;
    throw FlowErr{cat, msg};
}
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) raise(e == hipErrorOutOfMemory ? kOutOfMemory : kInternalError, "GpuError: %s: %s", what, hipGetErrorString(e));
}

// ---- JSON ------------------------------------------------------------------------------------------------------
struct JVal {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    double n = 0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* k) const {
        if (t != Obj) return nullptr;
        for (const auto& kv : o) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool is_null() const { return t == Null; }
};
struct JParser {
    const char *p, *end;
    int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    [[noreturn]] void bad(const char* what) { raise(kInvalidJson, "InvalidJson: %s at byte %ld", what, static_cast<long>(end - p)); }
    void lit(const char* w) { const size_t n = std::strlen(w); if (static_cast<size_t>(end - p) < n || std::memcmp(p, w, n)) bad("bad literal"); p += n; }
    std::string str() {
        std::string out;
        ++p;
        while (true) {
            if (p >= end) bad("unterminated string");
            const unsigned char c = static_cast<unsigned char>(*p++);
            if (c == '"') return out;
            if (c < 0x20) bad("control character in string");
            if (c != '\\') { out.push_back(static_cast<char>(c)); continue; }
            if (p >= end) bad("unterminated escape");
            const char e = *p++;
            switch (e) {
            case '"': case '\\': case '/': out.push_back(e); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                if (end - p < 4) bad("short \\u escape");
                unsigned v = 0;
                for (int i = 0; i < 4; ++i) {
                    const char h = *p++;
                    v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : (bad("bad hex digit"), 0));
                }
                if (v < 0x80) out.push_back(static_cast<char>(v));
                else if (v < 0x800) { out.push_back(static_cast<char>(0xC0 | (v >> 6))); out.push_back(static_cast<char>(0x80 | (v & 63))); }
                else { out.push_back(static_cast<char>(0xE0 | (v >> 12))); out.push_back(static_cast<char>(0x80 | ((v >> 6) & 63))); out.push_back(static_cast<char>(0x80 | (v & 63))); }
                break;
            }
            default: bad("bad escape");
            }
        }
    }
    JVal value() {
        if (++depth > 64) bad("nesting too deep");
        ws();
        if (p >= end) bad("unexpected end");
        JVal v;
        const char c = *p;
        if (c == '{') {
            v.t = JVal::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; --depth; return v; }
            while (true) {
                ws();
                if (p >= end || *p != '"') bad("expected a key");
                std::string k = str();
                ws();
                if (p >= end || *p != ':') bad("expected ':'");
                ++p;
                v.o.emplace_back(std::move(k), value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                bad("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.t = JVal::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; --depth; return v; }
            while (true) {
                v.a.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                bad("expected ',' or ']'");
            }
        } else if (c == '"') { v.t = JVal::Str; v.s = str(); }
        else if (c == 't') { lit("true"); v.t = JVal::Bool; v.b = true; }
        else if (c == 'f') { lit("false"); v.t = JVal::Bool; }
        else if (c == 'n') { lit("null"); }
        else {
            // RFC 8259 number grammar only: strtod alone would also take "inf", "nan", hex floats and leading '+'
            const char* q = p;
            if (q < end && *q == '-') ++q;
            if (q >= end || *q < '0' || *q > '9') bad("unexpected character");
            if (*q == '0') ++q; else while (q < end && *q >= '0' && *q <= '9') ++q;
            if (q < end && *q == '.') { ++q; if (q >= end || *q < '0' || *q > '9') bad("bad number"); while (q < end && *q >= '0' && *q <= '9') ++q; }
            if (q < end && (*q == 'e' || *q == 'E')) {
                ++q;
                if (q < end && (*q == '+' || *q == '-')) ++q;
                if (q >= end || *q < '0' || *q > '9') bad("bad number");
                while (q < end && *q >= '0' && *q <= '9') ++q;
            }
            if (q - p > 64) bad("number too long");
            const std::string tmp(p, static_cast<size_t>(q - p));
            v.n = std::strtod(tmp.c_str(), nullptr);
            if (!std::isfinite(v.n)) bad("number out of range");
            p = q;
            v.t = JVal::Num;
        }
        --depth;
        return v;
    }
};
JVal parse_json(const uint8_t* buf, size_t n) {
    JParser P{reinterpret_cast<const char*>(buf), reinterpret_cast<const char*>(buf) + n};
    JVal v = P.value();
    P.ws();
    if (P.p != P.end) P.bad("trailing characters");
    return v;
}
int64_t want_int(const JVal& o, const char* key, const char* node) {
    const JVal* v = o.get(key);
    if (!v || v->t != JVal::Num || v->n != std::floor(v->n) || std::fabs(v->n) > 9007199254740992.0)
        raise(kInvalidJson, "InvalidJson: %s.%s must be an integer", node, key);
    return static_cast<int64_t>(v->n);
}
uint32_t want_u32(const JVal& o, const char* key, const char* node) {
    const int64_t v = want_int(o, key, node);
    if (v < 0 || v > 0x7fffffff) raise(kInvalidJson, "InvalidJson: %s.%s out of range", node, key);
    return static_cast<uint32_t>(v);
}

// imageflow_types::Color (lib.rs:807-824) + imageflow_helpers/src/colors.rs:36-61 -> Color32 0xAARRGGBB
uint32_t parse_color(const JVal* v, const char* node) {
    if (!v || v->is_null()) return 0u;
    if (v->t == JVal::Str) {
        if (v->s == "transparent") return 0u;
        if (v->s == "black") return 0xFF000000u;
        raise(kInvalidJson, "InvalidJson: %s: unknown colour '%s'", node, v->s.c_str());
    }
    const JVal* srgb = v->get("srgb");
    const JVal* hex = srgb ? srgb->get("hex") : nullptr;
    if (!hex || hex->t != JVal::Str) raise(kInvalidJson, "InvalidJson: %s: colour must be \"transparent\", \"black\" or {\"srgb\":{\"hex\":..}}", node);
    std::string s = hex->s;
    if (!s.empty() && s[0] == '#') s.erase(0, 1);
    if (s.size() == 3 || s.size() == 4) { std::string d; for (char c : s) { d.push_back(c); d.push_back(c); } s = d; }
    if (s.size() == 6) s += "FF";
    if (s.size() != 8) raise(kArgumentInvalid, "InvalidNodeParams: %s: bad colour '%s'", node, hex->s.c_str());
    uint32_t ch[4];
    for (int i = 0; i < 4; ++i) {
        char* e = nullptr;
        const std::string part = s.substr(static_cast<size_t>(2 * i), 2);
        ch[i] = static_cast<uint32_t>(std::strtoul(part.c_str(), &e, 16));
        if (e != part.c_str() + 2) raise(kArgumentInvalid, "InvalidNodeParams: %s: bad colour '%s'", node, hex->s.c_str());
    }
    return (ch[3] << 24) | (ch[0] << 16) | (ch[1] << 8) | ch[2];
}
// s::Color::Transparent, the enum value (a missing colour reads as it): the only colour create_canvas.rs:79-82 turns into a
// ReplaceSelf canvas -- an srgb colour whose alpha is 0 still makes a BlendWithMatte canvas
bool keyword_transparent(const JVal* v) { return !v || v->is_null() || (v->t == JVal::Str && v->s == "transparent"); }
int parse_filter(const JVal* v, int dflt) {                      // imageflow_types/src/lib.rs:144-205
    if (!v || v->is_null()) return dflt;
    static const std::pair<const char*, int> names[] = {
        {"robidoux_fast", 1}, {"robidoux", 2}, {"robidoux_sharp", 3}, {"ginseng", 4}, {"ginseng_sharp", 5}, {"lanczos", 6},
        {"lanczos_sharp", 7}, {"lanczos_2", 8}, {"lanczos_2_sharp", 9}, {"cubic", 11}, {"cubic_sharp", 12}, {"catmull_rom", 13},
        {"mitchell", 14}, {"cubic_b_spline", 15}, {"hermite", 16}, {"jinc", 17}, {"triangle", 22}, {"linear", 23}, {"box", 24},
        {"fastest", 27}, {"n_cubic", 29}, {"n_cubic_sharp", 30}};
    if (v->t == JVal::Str)
        for (const auto& kv : names) if (v->s == kv.first) return kv.second;
    raise(kInvalidJson, "InvalidJson: unknown filter");
}

// `down.filter` / `up.filter` of a querystring: FilterStrings (imageflow_riapi/src/ir4/parsing.rs:159-193) spells every filter with
// and without underscores, any case; -> the JSON name parse_filter takes.  A value the reference would drop with a warning
// is refused here: a drop-in that cannot say "warning" must not pick another filter silently.
std::string querystring_filter_name(std::string v) {
    static const char* names[] = {"robidoux_fast", "robidoux", "robidoux_sharp", "ginseng", "ginseng_sharp", "lanczos", "lanczos_sharp", "lanczos_2",
                                  "lanczos_2_sharp", "cubic", "cubic_sharp", "catmull_rom", "mitchell", "cubic_b_spline", "hermite", "jinc", "triangle",
                                  "linear", "box", "fastest", "n_cubic", "n_cubic_sharp"};
    auto squash = [](std::string t) {
        std::string o;
        for (char ch : t) if (ch != '_') o.push_back(static_cast<char>(std::tolower(static_cast<unsigned char>(ch))));
        return o;
    };
    const std::string want = squash(v);
    for (const char* n : names) if (squash(n) == want) return n;
    raise(kArgumentInvalid, "InvalidNodeParams: querystring filter '%s' is not one of imageflow's filters", v.c_str());
}

// ---- frames in HBM ---------------------------------------------------------------------------------------------
// A decoded JPEG whose pixel stage has not run: when the frame's one consumer is a resample, decode and resample run as ONE
// device call (ifhip_jpeg_decode_resample_batch_device: no decoded BGRA bitmap in HBM); any other consumer gets the bitmap
// through Job::dev(), which runs the pixel stage then.
// ---- the stream of a job ------------------------------------------------------------------------------------------------
// One Context per thread (imageflow_abi/src/lib.rs:20-27): every job runs on a stream of its own (non-blocking: nothing a
// job does waits for another thread's job), leased from a pool for the duration of one send_json; device memory comes from
// the library's size-class cache (devmem.cpp) and goes back without a driver call.  Whatever a job frees it frees behind
// quiesce() -- a look at the job's stream, a wait if it is still busy -- so the block is idle (ifhip::QuiescedScope around
// the whole job); nodes themselves do not wait for the device.
std::mutex g_stream_mu;
struct DeviceQueues { std::vector<hipStream_t> pool; int slots_taken = 0; std::condition_variable slot_cv; };
std::map<int, DeviceQueues> g_queues;                     // per device ordinal: a stream belongs to the device it was created on
thread_local hipStream_t t_job_stream = nullptr;          // the stream of the job this thread is running (null outside a job)
// Admission: jobs beyond kJobSlots PER DEVICE wait for a slot on the host.  The rate grows with the jobs admitted -- batches
// of the coalesced decode fill up, the device always has somebody's pixel stage to run -- until the host's CPUs are the
// limit: 7 800 jobs/s at 20, 9 900 at 40, 11 100 at 48 for the thumbnail job on a 16-CPU container (round 5,
// profiles/r5_abi_jobs_slots_after_host_fixes.txt); 40 leaves a quarter of that container's CPUs to the caller.  (Rounds 4
// and 5 measured a FALL above 20 admitted jobs and blamed the runtime's hardware queues; it was the 17-32-file decode
// batches, see kMaxCoalesce.)
constexpr int kJobSlots = 40;
// Contexts and devices.  The reference's guidance is one Context per thread (imageflow_abi/src/lib.rs:20-27) and jobs are
// independent, so a node's GPUs are fed by giving every context a device: with ifhip_shim_spread_contexts(1) a new context
// takes the next usable device round-robin and all its jobs run there, whichever thread calls (no collective: a job's
// outputs are host buffers).  Without it (the default) a context follows the calling thread's current device, as before.
std::atomic<int> g_spread_contexts{0};
std::atomic<uint32_t> g_context_counter{0};
// HIP ordinals of the gfx950 devices, asked of the driver once (hipGetDeviceProperties fills a kilobyte struct per call, and
// a service creates thousands of contexts a second)
const std::vector<int>& usable_devices() {
    static const std::vector<int> list = [] {
        std::vector<int> v;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return v; }
        for (int i = 0; i < n; ++i) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) v.push_back(i);
            else (void)hipGetLastError();
        }
        return v;
    }();
    return list;
}
struct DeviceScope {                                      // the calling thread on the context's device for one call
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) {
            if (hipSetDevice(dev) == hipSuccess) switched = true;
            else (void)hipGetLastError();                 // (the first GPU call of the job reports)
        }
    }
    ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};
struct StreamLease {
    hipStream_t st = nullptr;
    int dev = 0;
    StreamLease() {
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        {
            std::unique_lock<std::mutex> lk(g_stream_mu);
            constexpr int slots = kJobSlots;
            DeviceQueues& q = g_queues[dev];              // (map nodes do not move: the reference survives the wait)
            while (q.slots_taken >= slots) q.slot_cv.wait(lk);
            ++q.slots_taken;
            if (!q.pool.empty()) { st = q.pool.back(); q.pool.pop_back(); }
        }
        if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }   // (no device: the null stream, the first GPU call reports)
        t_job_stream = st;
        ifhip_set_thread_stream(st);
    }
    ~StreamLease() {
        t_job_stream = nullptr;
        ifhip_set_thread_stream(nullptr);
        if (st) (void)static_cast<hipError_t>(ifhip::wait_stream(st));
        {
            std::lock_guard<std::mutex> lk(g_stream_mu);
            DeviceQueues& q = g_queues[dev];
            if (st) q.pool.push_back(st);
            --q.slots_taken;
            q.slot_cv.notify_one();                       // (one waiter of THIS device; waking all of them cost a third of the job rate at 64 threads)
        }
    }
};
// Before anything of a job goes back to the cache: the job's stream -- the only one its blocks were ever used on -- is idle.
void quiesce() {
    if (t_job_stream && hipStreamQuery(t_job_stream) != hipSuccess) (void)static_cast<hipError_t>(ifhip::wait_stream(t_job_stream));
    (void)hipGetLastError();
}
hipError_t job_malloc(void** p, size_t bytes) { return static_cast<hipError_t>(ifhip::cached_malloc(p, bytes)); }
void job_free(void* p) { if (p) { quiesce(); (void)ifhip::cached_free(p); } }

// The coefficient planes and quantisation tables of one entropy-decoded BATCH of files (one geometry): jobs of different
// threads whose decodes were coalesced into one device call each hold a share and read their own image's part.
struct DecodedBatch {
    int16_t* coef[3] = {nullptr, nullptr, nullptr};      // [n][bh_c][bw_c][64]
    size_t per_image[3] = {0, 0, 0};                     // int16 elements per image and component
    uint16_t* d_qt = nullptr;                            // [n][3][64]
    uint32_t w = 0, h = 0, bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
    int ncomp = 0;
    uint8_t hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1};
    // the last share goes when its job ends -- every job waits for its own stream before it lets go (quiesce), the
    // batch's decode itself was complete before any share was handed out
    // (... and on an error path between the decode's launch and its wait this destructor IS the wait: the planes may not
    // go back to the cache with the decode still writing them)
    ~DecodedBatch() {
        quiesce();
        for (int16_t* p : coef) if (p) (void)ifhip::cached_free(p);
        if (d_qt) (void)ifhip::cached_free(d_qt);
    }
};
struct PendingJpeg {
    ifhip_jpeg_stage* st = nullptr;
    int16_t* coef[3] = {nullptr, nullptr, nullptr};      // this image's planes inside `batch`
    uint16_t* d_qt = nullptr;
    std::shared_ptr<DecodedBatch> batch;
    ~PendingJpeg() {
        quiesce();
        if (st) ifhip_jpeg_stage_destroy(st);
        ifhip::QuiescedScope q;                          // (a job thread has one already; a share that dies elsewhere: see above)
        batch.reset();
    }
};
struct Frame {                                   // graphics/bitmaps.rs Bitmap: BGRA8, 64-byte row stride
    uint8_t* d = nullptr;                        // (null while `pending`: read it through Job::dev())
    uint32_t w = 0, h = 0, stride = 0;
    bool alpha = false;
    int compose = IFHIP_REPLACE_SELF;
    uint32_t matte = 0;
    std::unique_ptr<PendingJpeg> pending;
    size_t bytes() const { return static_cast<size_t>(h) * stride; }
    ~Frame() { job_free(d); }
};
using FramePtr = std::shared_ptr<Frame>;

// ExecutionSecurity (imageflow_types/src/lib.rs:1199-1236): the two limits that bound what this library allocates;
// defaults = sane_defaults(), a job may override them with `security` (v1/execute) / `builder_config.security` (v1/build)
struct SizeLimit { uint32_t w, h; float megapixels; };
struct Security {
    SizeLimit max_decode_size{12000, 12000, 100.f};
    SizeLimit max_frame_size{10000, 10000, 100.f};
};
// Engine::validate_frame_size (flow/execution_engine.rs:286-325); ErrorKind::SizeLimitExceeded -> ArgumentInvalid
void check_size(const SizeLimit& lim, const char* what, uint64_t w, uint64_t h) {
    if (w > lim.w) raise(kArgumentInvalid, "SizeLimitExceeded: Frame width %llu exceeds %s.w %u", static_cast<unsigned long long>(w), what, lim.w);
    if (h > lim.h) raise(kArgumentInvalid, "SizeLimitExceeded: Frame height %llu exceeds %s.h %u", static_cast<unsigned long long>(h), what, lim.h);
    const float mp = static_cast<float>(w) * static_cast<float>(h) / 1000000.f;
    if (mp > lim.megapixels) raise(kArgumentInvalid, "SizeLimitExceeded: Frame megapixels %f exceeds %s.megapixels %f", static_cast<double>(mp), what, static_cast<double>(lim.megapixels));
}

constexpr char kRawMagic[8] = {'I', 'F', 'B', 'G', 'R', 'A', '1', '\0'};
constexpr size_t kRawHeader = 8 + 16;

// ---- context -----------------------------------------------------------------------------------------------------
// Output buffers: Ready -> (get_output_buffer_by_id) Lent -> stays Lent; Ready -> (take_output_buffer) Taken
// (CodecInstanceContainer, imageflow_core/src/codecs/mod.rs:421-436,560-626)
enum class OutState { Ready, Lent, Taken };
struct Io {
    bool is_output = false;
    const uint8_t* in = nullptr;
    size_t in_len = 0;
    std::vector<uint8_t> owned;                  // copied inputs (lifetime_outlives_function_call) / output bytes
    bool written = false;
    OutState out_state = OutState::Ready;
    // v1/tell_decoder {jpeg_downscale_hints} (Context::tell_decoder -> MzDec::tell_decoder, mozjpeg_decoder.rs:560-586): kept
    // with the input until its decoder runs; a decode node's own `commands` are told after these and win
    bool out_base64 = false;                     // IoEnum::OutputBase64: the job result carries the bytes as {"base_64": ...}
    bool told = false;
    uint32_t told_w = 0, told_h = 0;
    bool told_spatial = false, told_gamma = false;
    bool told_discard_profile = false;           // DecoderCommand::DiscardColorProfile (mozjpeg_decoder.rs:88-91)
};
struct Response {
    int64_t status;
    std::string json;
};
}  // namespace

struct imageflow_json_response {                 // opaque to callers; read through imageflow_json_response_read
    Response r;
};
struct imageflow_context {
    std::mutex mu;
    std::map<int32_t, Io> io;
    int err_cat = kOk;
    std::string err_msg;
    std::vector<std::unique_ptr<imageflow_json_response>> responses;
    std::vector<std::unique_ptr<uint8_t[]>> allocations;
    // CancellationToken (imageflow_core/src/context.rs:50-131): a flag another thread may set while a job holds `mu`,
    // and the reference's debug-build poll countdown ("cancel at the n-th poll", :96-104) as a test hook
    std::atomic<bool> cancel{false};
    std::atomic<int64_t> poll_countdown{INT64_MAX};
    std::atomic<int64_t> fused_decode_resamples{0};
    std::atomic<int64_t> device_coded_files{0};          // JPEG outputs whose entropy coding ran on the device (diagnostic)
    std::atomic<int64_t> coalesced_decodes{0};           // decodes of this context that shared their device call with another thread's job
    std::atomic<int> device{-1};                         // device ordinal its jobs run on; -1: the calling thread's current device
    bool cancellation_requested() {
        if (cancel.load(std::memory_order_relaxed)) return true;
        if (poll_countdown.load(std::memory_order_relaxed) == INT64_MAX) return false;
        return poll_countdown.fetch_sub(1, std::memory_order_relaxed) < 1;
    }
    void set_error(int cat, const std::string& m) { if (err_cat == kOk) { err_cat = cat; err_msg = m; } }   // first error sticks
};

namespace {

std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(static_cast<char>(c)); }
        else if (c == '\n') o += "\\n";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back(static_cast<char>(c));
    }
    return o;
}
const imageflow_json_response* respond(imageflow_context* c, int64_t status, const std::string& json) {
    c->responses.emplace_back(new imageflow_json_response{Response{status, json}});
    return c->responses.back().get();
}
const imageflow_json_response* respond_error(imageflow_context* c, int cat, const std::string& msg) {
    c->set_error(cat, msg);                                      // imageflow_abi/src/lib.rs:1001-1008
    const int code = http_code(cat);                             // JsonResponse::fail_with_message, json/mod.rs:170-181
    return respond(c, code, "{\n  \"code\": " + std::to_string(code) + ",\n  \"success\": false,\n  \"message\": \"" +
                                json_escape(msg) + "\",\n  \"data\": \"none\"\n}");      // ResponsePayload::None, a unit variant renamed "none" (imageflow_types/src/lib.rs:2061-2062)
}

// ---- sizing: AspectRatio::proportional (imageflow_riapi/src/sizing.rs:118-185) ------------------------------------
// The other side of a box that keeps `sw x sh`'s ratio, snapping to the source's own side or to the requested box when
// the rounding loss of the requested box explains the difference (:83-116).
double rust_round(double v) { return std::round(v); }            // f64::round: half away from zero, as C's round()
int64_t proportional(int64_t sw, int64_t sh, int64_t basis, bool basis_is_width, bool have_target, int64_t tw, int64_t th) {
    const double ratio = static_cast<double>(sw) / static_cast<double>(sh);
    double snap_amount = 1.0 - 2.220446049250313e-16;
    if (have_target) {
        if (!basis_is_width) {                                    // rounding_loss_based_on_target_width (:83-98)
            const double recreate_y = static_cast<double>(sh) * (static_cast<double>(tw) / static_cast<double>(sw));
            snap_amount = std::fabs(static_cast<double>(tw) - rust_round(recreate_y) * ratio);
        } else {                                                  // rounding_loss_based_on_target_height (:99-115)
            const double recreate_x = static_cast<double>(sw) * (static_cast<double>(th) / static_cast<double>(sh));
            snap_amount = std::fabs(static_cast<double>(th) - rust_round(recreate_x) / ratio);
        }
    }
    const int64_t snap_a = basis_is_width ? sh : sw;
    const int64_t snap_b = have_target ? (basis_is_width ? th : tw) : snap_a;
    const double f = basis_is_width ? static_cast<double>(basis) / ratio : ratio * static_cast<double>(basis);
    const double da = std::fabs(f - static_cast<double>(snap_a)), db = std::fabs(f - static_cast<double>(snap_b));
    int64_t v;
    if (da <= snap_amount && da <= db) v = snap_a;
    else if (db <= snap_amount) v = snap_b;
    else {
        const double r = rust_round(f);
        if (r <= -2147483648.0 || r >= 2147483647.0) raise(kArgumentInvalid, "LayoutError: ValueScalingFailed");
        v = static_cast<int64_t>(r);
    }
    if (v < 0) raise(kArgumentInvalid, "LayoutError: ValueScalingFailed");
    return v == 0 ? 1 : v;
}
// AspectRatio::box_of(target, Inner) (:189-197): the largest sw:sh box inside tw x th
void inner_box(int64_t sw, int64_t sh, int64_t tw, int64_t th, int64_t* ow, int64_t* oh) {
    const double rs = static_cast<double>(sw) / static_cast<double>(sh), rt = static_cast<double>(tw) / static_cast<double>(th);
    if (rs > rt) { *ow = tw; *oh = proportional(sw, sh, tw, true, true, tw, th); }
    else { *ow = proportional(sw, sh, th, false, true, tw, th); *oh = th; }
}

// ---- the job interpreter ---------------------------------------------------------------------------------------
struct EncodeRecord { int32_t io_id; uint32_t w, h; const char* mime; const char* ext; };
struct DecodeRecord { int32_t io_id; uint32_t w, h; const char* mime; const char* ext; };
struct NodePerf { const char* name; uint64_t wall_ns; float gpu_ms; hipEvent_t e0, e1; };   // s::NodePerf (imageflow_types/src/lib.rs:1999-2002); e0/e1: read at the job's end
// The hipEvents behind a node's `gpu_microseconds`, kept between jobs (two driver calls per node saved) and per device.
struct TimingEvents {
    std::mutex mu;
    std::map<int, std::vector<hipEvent_t>> spare;
    hipEvent_t take(int device) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = spare[device];
            if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    }
    void give(int device, hipEvent_t e) {
        if (!e) return;
        std::lock_guard<std::mutex> lk(mu);
        auto& v = spare[device];
        if (v.size() < 256) v.push_back(e); else (void)hipEventDestroy(e);
    }
};
TimingEvents& timing_events() { static TimingEvents* t = new TimingEvents; return *t; }   // (never destroyed: jobs may outlive static teardown)
struct ResampleHints {                                                        // s::ResampleHints (lib.rs:925-933), parsed
    bool has_sharpen = false;
    float sharpen = 0.f;
    int down = IFHIP_FILTER_ROBIDOUX, up = IFHIP_FILTER_GINSENG, space = IFHIP_SPACE_LINEAR;   // scale_render.rs:257-261,278-279
    bool has_bg = false, bg_keyword_transparent = false;
    uint32_t bg = 0;
    enum When { kDefault, kSizeDiffers, kSizeDiffersOrSharpen, kAlways } resample_when = kDefault;
    enum SWhen { kSAlways, kSDown, kSUp, kSSizeDiffers } sharpen_when = kSAlways;
};

// ---- decodes of different threads as ONE device call ---------------------------------------------------------------------
// A job decodes one file: 14 workgroups of the entropy kernels for a 4K JPEG, on a chip of 256 CUs, and the HIP runtime
// runs the streams of a process on a handful of hardware queues -- measured through this ABI (round 4): 5 300 jobs/s at 16
// threads and FEWER beyond, the GPU a fifth busy.  The entropy stage takes a batch of files of one geometry as cheaply as
// one (ifhip_jpeg_entropy_create / _decode_device), so concurrent decodes are coalesced: the first thread to arrive leads,
// gives the others a moment (only while other jobs are in flight at all), decodes everybody's file of its geometry in one
// call on its own stream and hands each job its image's planes; the rest wait on a condition variable.  A batch that fails
// (one damaged file) sends every member back to decode alone, so errors stay with the job that owns them.
// At most 16 files to a batch: the decode's cost per file is flat from 8 up (profiles/r5_abi_jobs_cliff_7_decode_batches.txt),
// and batches of 17-32 -- a sixth size class of 130 MB coefficient planes, everybody's pixel stages released at once --
// took 20-35 ms each where 16 take 3: the job rate fell from 6 500 to 800 whenever more than ~20 jobs were admitted
// (round 5, profiles/r5_abi_jobs_cliff_6_batch_cap.txt).
constexpr uint32_t kMaxCoalesce = 16;
std::atomic<int> g_jobs_in_flight{0};
struct DecodeRequest {
    ifhip_jpeg_prepared* prepared = nullptr;             // the job's file, prepared on the job's own thread
    uint32_t w = 0, h = 0;
    int ncomp = 0;
    uint8_t hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
    // result
    std::shared_ptr<DecodedBatch> batch;
    uint32_t index = 0, batch_size = 0;
    bool done = false, retry_alone = false, taken = false;      // taken: a leader is decoding it
    std::condition_variable cv;                                 // this request's own wake-up (never a broadcast to every waiting job)
    bool same_geometry(const DecodeRequest& o) const {
        return w == o.w && h == o.h && ncomp == o.ncomp && std::memcmp(hs, o.hs, 3) == 0 && std::memcmp(vs, o.vs, 3) == 0;
    }
};
// one batch on the calling thread's job stream; throws FlowErr
std::shared_ptr<DecodedBatch> decode_files(const std::vector<DecodeRequest*>& reqs) {
    const uint32_t n = static_cast<uint32_t>(reqs.size());
    std::vector<ifhip_jpeg_prepared*> files(n);
    for (uint32_t i = 0; i < n; ++i) files[i] = reqs[i]->prepared;
    ifhip_jpeg_entropy* ent = nullptr;
    check(ifhip_jpeg_entropy_create_prepared(&ent, files.data(), n));
    struct EntGuard { ifhip_jpeg_entropy* e; ~EntGuard() { quiesce(); ifhip_jpeg_entropy_destroy(e); } } eg{ent};
    auto b = std::make_shared<DecodedBatch>();
    uint32_t nsub = 0, nseg = 0;
    check(ifhip_jpeg_entropy_info(ent, &b->w, &b->h, &b->ncomp, b->hs, b->vs, b->bw, b->bh, &nsub, &nseg));
    uint32_t n_cap = 1;                                          // (batch sizes as powers of two: six size classes in the cache, not thirty-two)
    while (n_cap < n) n_cap *= 2;
    for (int k = 0; k < 3; ++k) {
        b->per_image[k] = static_cast<size_t>(b->bw[k]) * b->bh[k] * 64u;
        hip_check(job_malloc(reinterpret_cast<void**>(&b->coef[k]), std::max<size_t>(1, b->per_image[k] * n_cap) * 2u), "hipMalloc(coefficients)");
    }
    // the quantisation tables ride along: queued before the decode, whose own wait for the stream covers them
    std::vector<uint16_t> qt(static_cast<size_t>(n) * 192u);
    check(ifhip_jpeg_entropy_quant_tables(ent, qt.data()));
    hip_check(job_malloc(reinterpret_cast<void**>(&b->d_qt), qt.size() * 2u), "hipMalloc(qt)");
    struct PinGuard { void* p = nullptr; ~PinGuard() { if (p) { quiesce(); (void)ifhip::cached_host_free(p); } } } qt_pin;
    hip_check(static_cast<hipError_t>(ifhip::stage_to_device(b->d_qt, qt.data(), qt.size() * 2u, &qt_pin.p)), "upload(qt)");
    uint32_t rounds = 0;
    check(ifhip_jpeg_entropy_decode_device(ent, b->coef[0], b->coef[1], b->coef[2], &rounds, t_job_stream));
    return b;
}
struct DecodeCoalescer {
    std::mutex mu;
    std::condition_variable leader_cv;                              // the gathering leader: "somebody arrived"
    std::vector<DecodeRequest*> queue;                              // requests nobody has taken yet, in arrival order
    bool leader_active = false;
    int decoding = 0;                                               // batches whose decode is on the device right now
    // Wake-ups are targeted: a request sleeps on its OWN condition variable and is woken when its batch is done or when it
    // is its turn to lead; arrivals wake only a leader that is counting them.  (One shared condition variable with
    // notify_all woke every waiting job on every arrival and every hand-over.  It was NOT what made the job rate fall above
    // 20 admitted jobs -- profiles/r5_abi_jobs_cliff_1_targeted_wakeups_no_effect.txt; that was the batch size, kMaxCoalesce --
    // but forty sleepers woken for one hand-over is work nobody needs.)
    void wake_next_leader() {                                       // (mu held) the oldest request nobody took: it may lead now
        if (!leader_active && !queue.empty()) queue.front()->cv.notify_one();
    }
    // -> r.batch / r.index set, or r.retry_alone
    void submit(DecodeRequest& r) {
        std::unique_lock<std::mutex> lk(mu);
        queue.push_back(&r);
        if (leader_active) leader_cv.notify_one();
        for (;;) {
            // At most `max_decoding` batches decode at a time; whoever arrives meanwhile waits here, and the next leader takes
            // ALL of them: the batch size follows the load by itself (group commit).  An entropy decode costs about the same
            // half millisecond for 1 file or 16 (three latency-bound launches), so a batch of 8 is an eighth of
            // the device time per job -- before, a new batch formed as soon as the last one had been gathered, 1.85 files per
            // batch at 2 300 jobs/s (round 5, profiles/r5_abi_trace_cfg4_8_threads.txt).
            constexpr int max_decoding = 2;
            while (!r.done && (r.taken || leader_active || decoding >= max_decoding || queue.front() != &r)) r.cv.wait(lk);
            if (r.done) return;
            leader_active = true;                                    // the oldest untaken request leads, for one batch
            // the moment given to the others: only while other jobs are in flight at all (a lone caller pays nothing) and no
            // decode is running (else the wait above was the moment)
            long window_us = (g_jobs_in_flight.load(std::memory_order_relaxed) > 1 && decoding == 0) ? 120 : 0;
            size_t wait_for = kMaxCoalesce;
            if (const char* e = ifhip::debug_switch("coalesce_window_us")) window_us = std::atol(e);
            if (const char* e = ifhip::debug_switch("coalesce_wait_for")) wait_for = static_cast<size_t>(std::max(1L, std::atol(e)));
            if (window_us > 0) {
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
                while (queue.size() < wait_for && leader_cv.wait_until(lk, until) != std::cv_status::timeout) {}
            }
            // this thread's own request first, then whoever shares its geometry, in arrival order
            // Whatever leaves this block early (a bad_alloc while the lists are built) hands the leadership on and takes this
            // request out of the queue: a leader flag left set would park every later job of the device on its own condition
            // variable for good, and the request itself dies with its caller's frame.
            struct Leading {
                DecodeCoalescer& c; DecodeRequest* me; bool armed = true;
                ~Leading() {                                         // (mu held on every path that gets here armed)
                    if (!armed) return;
                    c.queue.erase(std::remove(c.queue.begin(), c.queue.end(), me), c.queue.end());
                    c.leader_active = false;
                    c.wake_next_leader();
                }
            } leading{*this, &r};
            std::vector<DecodeRequest*> mine{&r}, rest;
            for (DecodeRequest* q : queue)
                if (q != &r) (mine.size() < kMaxCoalesce && q->same_geometry(r) ? mine : rest).push_back(q);
            queue.swap(rest);
            for (DecodeRequest* q : mine) q->taken = true;
            leading.armed = false;
            leader_active = false;                                   // leading = gathering: the next batch forms while this one decodes
            ++decoding;
            if (decoding < max_decoding) wake_next_leader();
            lk.unlock();
            std::shared_ptr<DecodedBatch> b;
            bool failed = false;
            try { b = decode_files(mine); } catch (...) { failed = true; }      // (anything at all: the batch's members retry alone and report their own error)
            lk.lock();
            --decoding;
            for (size_t i = 0; i < mine.size(); ++i) {
                DecodeRequest* q = mine[i];
                if (failed) q->retry_alone = true;
                else { q->batch = b; q->index = static_cast<uint32_t>(i); q->batch_size = static_cast<uint32_t>(mine.size()); }
                q->done = true;
                if (q != &r) q->cv.notify_one();
            }
            wake_next_leader();
        }
    }
};
DecodeCoalescer& coalescer_for_device() {
    static std::mutex mu;
    static std::map<int, DecodeCoalescer*> per_device;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    DecodeCoalescer*& c = per_device[dev];
    if (!c) c = new DecodeCoalescer;
    return *c;
}

// Resample plans (contribution tables of one shape on the device, immutable, thread-safe) are shared by all jobs of the
// process: a service resizes to a handful of sizes, and a plan costs a dozen uploads.  Least recently used of 256 goes.
struct PlanKey {
    int device; uint32_t in_w, in_h, w, h; int filter; uint32_t sharpen_bits;
    bool operator<(const PlanKey& o) const {
        return std::tie(device, in_w, in_h, w, h, filter, sharpen_bits) < std::tie(o.device, o.in_w, o.in_h, o.w, o.h, o.filter, o.sharpen_bits);
    }
};
std::mutex g_plan_mu;
typedef std::map<PlanKey, std::pair<std::shared_ptr<ifhip_resample_plan>, uint64_t>> PlanMap;
PlanMap& plan_map() { static PlanMap* m = new PlanMap; return *m; }        // never destroyed: no HIP calls from static destructors at exit
uint64_t g_plan_clock = 0;
std::shared_ptr<ifhip_resample_plan> shared_plan(uint32_t in_w, uint32_t in_h, uint32_t w, uint32_t h, int filter, float sharpen) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t bits;
    std::memcpy(&bits, &sharpen, 4);
    const PlanKey key{dev, in_w, in_h, w, h, filter, bits};
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = plan_map().find(key);
        if (it != plan_map().end()) { it->second.second = ++g_plan_clock; return it->second.first; }
    }
    ifhip_resample_plan* raw = nullptr;
    check(ifhip_resample_plan_create(&raw, in_w, in_h, w, h, filter, sharpen));
    // (a plan dropped from the cache while a job still holds it is destroyed by that job's thread, behind its stream's wait;
    // one dropped with no holder was last used by a job that has ended: plain destroy)
    // A plan's tables are only ever READ by kernels, and every job waits for its stream before it drops its reference
    // (PendingJpeg / Job teardown), so the last reference -- whoever holds it -- goes with nothing in flight on the plan; the
    // deleter still waits for the releasing thread's stream and, outside a job, the whole device (cached_free).
    std::shared_ptr<ifhip_resample_plan> sp(raw, [](ifhip_resample_plan* q) { quiesce(); ifhip_resample_plan_destroy(q); });
    std::shared_ptr<ifhip_resample_plan> evicted;                // released AFTER the lock: its deleter waits for a stream
    std::shared_ptr<ifhip_resample_plan> result;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        PlanMap& plans = plan_map();
        if (plans.size() >= 256) {
            auto victim = plans.begin();
            for (auto it = plans.begin(); it != plans.end(); ++it) if (it->second.second < victim->second.second) victim = it;
            evicted = std::move(victim->second.first);
            plans.erase(victim);
        }
        auto ins = plans.emplace(key, std::make_pair(sp, ++g_plan_clock));
        result = ins.first->second.first;
    }
    return result;
}

struct Job {
    imageflow_context* c;
    std::vector<EncodeRecord> encodes;
    std::vector<DecodeRecord> decodes;
    std::vector<NodePerf> perf;
    Security sec;
    std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();

    // return_if_cancelled! (imageflow_core/src/errors.rs:124-140; polled in Engine::graph_execute before every node,
    // execution_engine.rs:502, at every bitmap borrow, context.rs:389-401, and inside the codecs)
    void poll_cancel() {
        if (c->cancellation_requested()) raise(kOperationCancelled, "OperationCancelled: the job was cancelled");
    }

    // one executed primitive: wall clock as the reference's per-node cost (execution_engine.rs:506-538) plus the time
    // the device spent on it (hipEvents on the stream every node of this interpreter launches on)
    // A node's performance record: wall time on the host and, between two hipEvents on the job's stream, the device's time.
    // The node does NOT wait for the device when it ends -- its launches stay in flight while the host prepares the next
    // node's; the events are read once, at the job's end (settle_perf).  (Until round 5 every node ended in
    // hipEventSynchronize: 4 host/device round trips per job and, under the runtime's default spinning wait, one busy host
    // core per job in flight -- profiles/r5_abi_jobs_host_cpu.txt.)
    struct Timed {
        Job* j; const char* name; std::chrono::steady_clock::time_point t0; hipEvent_t e0 = nullptr, e1 = nullptr;
        Timed(Job* job, const char* n) : j(job), name(n), t0(std::chrono::steady_clock::now()) {
            const int dev = j->c->device.load(std::memory_order_relaxed);
            e0 = timing_events().take(dev); e1 = timing_events().take(dev);
            if (!e0 || !e1 || hipEventRecord(e0, t_job_stream) != hipSuccess) {
                (void)hipGetLastError();
                timing_events().give(dev, e0); timing_events().give(dev, e1);
                e0 = e1 = nullptr;
            }
        }
        ~Timed() {
            if (e0 && hipEventRecord(e1, t_job_stream) != hipSuccess) {
                (void)hipGetLastError();
                const int dev = j->c->device.load(std::memory_order_relaxed);
                timing_events().give(dev, e0); timing_events().give(dev, e1);
                e0 = e1 = nullptr;
            }
            const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            try { j->perf.push_back({name, static_cast<uint64_t>(ns), 0.f, e0, e1}); }
            catch (...) { const int dev = j->c->device.load(std::memory_order_relaxed); timing_events().give(dev, e0); timing_events().give(dev, e1); }
        }
    };
    // the device times of the nodes, once the job's stream has drained; `read` false (a failed job): the events just go back
    void settle_perf(bool read) {
        bool any = false;
        for (const NodePerf& n : perf) any = any || n.e0;
        if (!any) return;
        const bool drained = read && ifhip::wait_stream(t_job_stream) == 0;
        const int dev = c->device.load(std::memory_order_relaxed);
        for (NodePerf& n : perf) {
            if (!n.e0) continue;
            if (drained && hipEventElapsedTime(&n.gpu_ms, n.e0, n.e1) != hipSuccess) { (void)hipGetLastError(); n.gpu_ms = 0.f; }
            timing_events().give(dev, n.e0); timing_events().give(dev, n.e1);
            n.e0 = n.e1 = nullptr;
        }
    }
    ~Job() { settle_perf(false); }

    // matte_canvas: the frame is CreateCanvas' result for a colour other than the enum value Transparent (-1: decide by the
    // colour's alpha, for callers that have no JSON colour)
    FramePtr new_frame(uint32_t w, uint32_t h, bool alpha, uint32_t fill_color32 = 0, bool zero = true, int matte_canvas = -1) {
        if (w == 0 || h == 0) raise(kArgumentInvalid, "InvalidArgument: Bitmap dimensions cannot be zero");
        check_size(sec.max_frame_size, "max_frame_size", w, h);
        poll_cancel();                                               // borrow_bitmaps_mut (context.rs:389)
        auto f = std::make_shared<Frame>();
        f->w = w; f->h = h; f->stride = ifhip_stride_for_width(w); f->alpha = alpha;
        if (f->stride == 0) raise(kArgumentInvalid, "InvalidArgument: Bitmap width %u has no 32-bit stride", w);
        hip_check(job_malloc(reinterpret_cast<void**>(&f->d), f->bytes() + 64), "hipMalloc(frame)");
        if (zero) hip_check(hipMemsetAsync(f->d, 0, f->bytes() + 64, t_job_stream), "hipMemset(frame)");
        if (matte_canvas < 0 ? (fill_color32 >> 24) != 0 : matte_canvas != 0) { f->compose = IFHIP_BLEND_WITH_MATTE; f->matte = fill_color32; }   // create_canvas.rs:77-103
        if (fill_color32 >> 24) {                    // bitmaps.rs:829-837: canvases start filled unless the colour is transparent
            check(ifhip_fill_rect_batch_device(f->d, f->bytes(), 1, w, h, f->stride, IFHIP_REPLACE_SELF, 0, 0, w, h, fill_color32, t_job_stream));
        }
        return f;
    }
    // the frame's pixels on the device; a pending JPEG gets its pixel stage now (IDCT + colour into a bitmap)
    uint8_t* dev(const FramePtr& f) {
        if (f->pending) {
            uint8_t* d = nullptr;                                    // the frame takes the bitmap only once it holds the pixels:
            hip_check(job_malloc(reinterpret_cast<void**>(&d), f->bytes() + 64), "hipMalloc(frame)");     // a failed stage leaves
            struct Guard { uint8_t* p; ~Guard() { job_free(p); } } g{d};                                  // `pending` and no buffer
            PendingJpeg& p = *f->pending;
            check(ifhip_jpeg_idct_color_batch_device(p.st, p.coef[0], p.coef[1], p.coef[2], p.d_qt, 1, d, f->bytes(), f->stride, t_job_stream));
            hip_check(static_cast<hipError_t>(ifhip::wait_stream(t_job_stream)), "decode");
            f->d = d; g.p = nullptr;
            f->pending.reset();
        }
        return f->d;
    }
    bool lazy_decode = false;                                        // the decode node being run has exactly one consumer
    // clone_node: the Clone node MutProtect puts before a mutating node whose parent has other children (definitions.rs:
    // 320-341) = CreateCanvas(Transparent, parent.fmt) + CopyRectToCanvas (clone_crop_fill_expand.rs:151-175), which leaves
    // the copy in BlendWithSelf (copy_rect.rs:37); otherwise a private copy of this interpreter's that keeps the frame's state
    FramePtr clone(const FramePtr& in, bool clone_node = false) {
        FramePtr c2 = new_frame(in->w, in->h, in->alpha, 0, false);
        hip_check(hipMemcpyAsync(c2->d, dev(in), in->bytes(), hipMemcpyDeviceToDevice, t_job_stream), "clone");
        if (clone_node) c2->compose = IFHIP_BLEND_WITH_SELF;
        else { c2->compose = in->compose; c2->matte = in->matte; }
        return c2;
    }

    Io& input(int32_t id) {
        auto it = c->io.find(id);
        if (it == c->io.end() || it->second.is_output) raise(kArgumentInvalid, "InvalidArgument: io_id %d is not a registered input", id);
        return it->second;
    }
    Io& output(int32_t id) {
        auto it = c->io.find(id);
        if (it == c->io.end() || !it->second.is_output) raise(kArgumentInvalid, "InvalidArgument: io_id %d is not a registered output", id);
        return it->second;
    }

    // MzDec::get_exif_rotation_flag (mozjpeg_decoder.rs:290-292): the EXIF orientation tag of a JPEG input, -1 = none
    static int exif_flag(const Io& in) {
        int flag = -1;
        if (in.in_len >= 4 && in.in[0] == 0xFF && in.in[1] == 0xD8) (void)ifhip_jpeg_exif_orientation(in.in, in.in_len, &flag);
        return flag;
    }
    // header facts of an input (get_scaled_rotated_image_info's part that watermark / command_string need); `rotated`:
    // Context::swap_dimensions_by_exif (context.rs:486-501) -- flags 5..8 swap width and height
    void image_size(int32_t io_id, uint32_t* w, uint32_t* h, bool rotated = false) {
        Io& in = input(io_id);
        if (rotated) {
            image_size(io_id, w, h, false);
            const int flag = exif_flag(in);
            if (flag >= 5 && flag <= 8) std::swap(*w, *h);
            return;
        }
        if (in.in_len >= kRawHeader && std::memcmp(in.in, kRawMagic, 8) == 0) {
            uint32_t hdr[4];
            std::memcpy(hdr, in.in + 8, 16);
            *w = hdr[0]; *h = hdr[1];
            return;
        }
        int nc = 0;
        uint8_t hs[3], vs[3];
        uint32_t bw[3], bh[3], ri = 0;
        uint16_t qt[192];
        if (in.in_len < 3 || in.in[0] != 0xFF || in.in[1] != 0xD8)
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: io_id %d is neither a JPEG nor the raw BGRA extension", io_id);
        int progressive = 0;
        (void)ri;
        const int rc = ifhip_jpeg_frame_info(in.in, in.in_len, w, h, &nc, hs, vs, bw, bh, qt, &progressive);
        if (rc == IFHIP_METHOD_NOT_IMPLEMENTED)
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: %s (Huffman-coded 8-bit JPEG with 1 or 3 components only; keep other files on libjpeg)", ifhip_last_error_message());
        check(rc);
    }

    // Progressive / multi-scan files: entropy decoding on the host (csrc/jpeg_read.cpp), planes uploaded, everything else as usual
    std::shared_ptr<DecodedBatch> decode_on_host(const Io& in) {
        auto b = std::make_shared<DecodedBatch>();
        uint16_t qt[192];
        int progressive = 0;
        const int frc = ifhip_jpeg_frame_info(in.in, in.in_len, &b->w, &b->h, &b->ncomp, b->hs, b->vs, b->bw, b->bh, qt, &progressive);
        if (frc == IFHIP_METHOD_NOT_IMPLEMENTED)
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: %s (Huffman-coded 8-bit JPEG with 1 or 3 components only; keep other files on libjpeg)", ifhip_last_error_message());
        check(frc);
        std::vector<int16_t> host[3];
        for (int k = 0; k < 3; ++k) {
            b->per_image[k] = static_cast<size_t>(b->bw[k]) * b->bh[k] * 64u;
            host[k].resize(std::max<size_t>(b->per_image[k], 64u));
        }
        poll_cancel();
        check(ifhip_jpeg_read_coefficients_host(in.in, in.in_len, host[0].data(), host[1].data(), host[2].data(), qt));
        for (int k = 0; k < 3; ++k) {
            hip_check(job_malloc(reinterpret_cast<void**>(&b->coef[k]), host[k].size() * 2u), "hipMalloc(coefficients)");
            hip_check(static_cast<hipError_t>(ifhip::copy_to_device(b->coef[k], host[k].data(), host[k].size() * 2u)), "upload(coefficients)");
        }
        hip_check(job_malloc(reinterpret_cast<void**>(&b->d_qt), sizeof qt), "hipMalloc(qt)");
        hip_check(static_cast<hipError_t>(ifhip::copy_to_device(b->d_qt, qt, sizeof qt)), "upload(qt)");
        return b;
    }

    // decode: MozJpegDecoder::read_frame (codecs/mozjpeg_decoder.rs:295-420) on the device, or the raw extension
    FramePtr decode(int32_t io_id, uint32_t hint_w, uint32_t hint_h, bool luma_spatial, bool luma_srgb) {
        Timed t(this, "primitive_decoder");
        Io& in = input(io_id);
        if (in.in_len >= kRawHeader && std::memcmp(in.in, kRawMagic, 8) == 0) {
            uint32_t hdr[4];
            std::memcpy(hdr, in.in + 8, 16);
            const uint32_t w = hdr[0], h = hdr[1], stride = hdr[2];
            if (w == 0 || h == 0 || stride < w * 4ull || (stride & 3u) || in.in_len < kRawHeader + static_cast<size_t>(h - 1) * stride + w * 4ull)
                raise(kImageMalformed, "ImageMalformed: raw BGRA container header does not match its length");
            check_size(sec.max_decode_size, "max_decode_size", w, h);
            FramePtr f = new_frame(w, h, hdr[3] != 0, 0, true);
            hip_check(hipMemcpy2DAsync(f->d, f->stride, in.in + kRawHeader, stride, w * 4ull, h, hipMemcpyHostToDevice, t_job_stream), "upload(raw frame)");
            hip_check(static_cast<hipError_t>(ifhip::wait_stream(t_job_stream)), "upload(raw frame)");
            decodes.push_back({io_id, w, h, "application/x-imageflow-bgra", "ifbgra"});
            return f;
        }
        if (in.in_len < 3 || in.in[0] != 0xFF || in.in[1] != 0xD8)                        // codecs/mod.rs:398-415 sniffing
            raise(kImageTypeNotSupported, "ImageTypeNotSupported: io_id %d is neither a JPEG nor the raw BGRA extension", io_id);
        {   // limits before anything is staged (mozjpeg_decoder.rs:196-214 checks max_decode_size on the header)
            uint32_t hw = 0, hh = 0;
            image_size(io_id, &hw, &hh);
            check_size(sec.max_decode_size, "max_decode_size", hw, hh);
        }
        {   // MzDec::read_frame transforms the frame to sRGB whenever the file carries an ICC profile (mozjpeg_decoder.rs:370-420)
            // unless the decoder was told discard_color_profile (:88-91).  Colour management is not part of this library: a
            // profile that is not sRGB itself would come out with other colours than the reference's, so the job is refused --
            // loudly -- instead (a profile that IS sRGB: the reference's transform is the identity up to its rounding).
            int kind = 0;
            if (!in.told_discard_profile && ifhip_jpeg_icc_profile_kind(in.in, in.in_len, &kind) == IFHIP_OK && kind == 2)
                raise(kActionNotSupported, "ActionNotSupported: io_id %d carries an embedded ICC profile that is not sRGB; this build has no colour "
                      "management (the reference converts such frames to sRGB, codecs/mozjpeg_decoder.rs:409).  Tell the decoder "
                      "\"discard_color_profile\" to decode the samples as they are, or keep the file on the reference.", io_id);
        }
        // the entropy stage: this file, together with whatever other threads' jobs want decoded right now (DecodeCoalescer)
        DecodeRequest rq;
        const int prc = ifhip_jpeg_entropy_prepare(&rq.prepared, in.in, in.in_len);          // parse, un-stuff, pack, tables: on this job's thread
        struct PreparedGuard { ifhip_jpeg_prepared* p; ~PreparedGuard() { ifhip_jpeg_prepared_destroy(p); } } prepared_guard{rq.prepared};
        if (prc == IFHIP_METHOD_NOT_IMPLEMENTED) {
            // not a single-scan baseline file: progressive (what the reference's mozjpeg preset writes) or multi-scan -- the scans
            // are decoded on the host as MzDec does (libjpeg's jdphuff.c), the pixel stage behind them stays on the GPU
            rq.batch = decode_on_host(in);
            rq.index = 0; rq.batch_size = 1;
        } else {
            check(prc);
            check(ifhip_jpeg_prepared_info(rq.prepared, &rq.w, &rq.h, &rq.ncomp, rq.hs, rq.vs));
            check(ifhip_jpeg_prepared_upload(rq.prepared, t_job_stream));      // the file's PCIe transfer starts now, not when a batch leader gets to it
            poll_cancel();                                           // the decoder's cancellation point (mozjpeg_decoder.rs:346-362 loop)
            coalescer_for_device().submit(rq);
            if (rq.retry_alone) {                                    // the shared call failed (somebody's file, maybe this one): alone, errors are this job's
                std::vector<DecodeRequest*> one{&rq};
                rq.batch = decode_files(one);
                rq.index = 0; rq.batch_size = 1;
            }
        }
        if (rq.batch_size > 1) c->coalesced_decodes.fetch_add(1, std::memory_order_relaxed);
        const std::shared_ptr<DecodedBatch> batch = rq.batch;
        const uint32_t w = batch->w, h = batch->h;
        const int ncomp = batch->ncomp;
        const uint8_t* hs = batch->hs;
        const uint8_t* vs = batch->vs;
        // MzDec::apply_downscaling (mozjpeg_decoder.rs:588-618): smallest i/8 (7 skipped) that still covers the hint
        int scale = 8;
        if (hint_w > 0 && hint_h > 0 && (w > hint_w || h > hint_h))                   // downscale_if_wider_than = width, or_if_taller_than = height (:72-77)
            for (int i = 1; i < 8; ++i) {
                if (i == 7) continue;
                if ((static_cast<uint64_t>(w) * i + 7) / 8 >= hint_w && (static_cast<uint64_t>(h) * i + 7) / 8 >= hint_h) { scale = i; break; }
            }
        ifhip_jpeg_stage* st = nullptr;
        const bool spatial = scale < 8 && luma_spatial;
        check(ifhip_jpeg_stage_create(&st, w, h, ncomp, hs, vs, scale, spatial ? 1 : 0, spatial && luma_srgb ? 1 : 0, 1));
        auto pend = std::make_unique<PendingJpeg>();                                  // owns the stage and a share of the decoded batch
        pend->st = st;
        pend->batch = batch;
        for (int k = 0; k < 3; ++k) pend->coef[k] = batch->coef[k] + static_cast<size_t>(rq.index) * batch->per_image[k];
        pend->d_qt = batch->d_qt + static_cast<size_t>(rq.index) * 192u;
        uint32_t ow = 0, oh = 0;
        check(ifhip_jpeg_stage_output_size(st, &ow, &oh));
        if (ow == 0 || oh == 0) raise(kArgumentInvalid, "InvalidArgument: Bitmap dimensions cannot be zero");
        check_size(sec.max_frame_size, "max_frame_size", ow, oh);
        poll_cancel();                                                                // borrow_bitmaps_mut (context.rs:389), as new_frame
        auto f = std::make_shared<Frame>();                                           // alpha not meaningful (:101-123)
        f->w = ow; f->h = oh; f->stride = ifhip_stride_for_width(ow); f->alpha = false;
        f->pending = std::move(pend);
        if (!lazy_decode) (void)dev(f);                                               // several consumers: one bitmap for all of them
        {   // Context::get_image_decodes reports get_unscaled_rotated_image_info (context.rs:519-538)
            const int flag = exif_flag(in);
            const bool swap = flag >= 5 && flag <= 8;
            decodes.push_back({io_id, swap ? h : w, swap ? w : h, "image/jpeg", "jpg"});
        }
        return f;
    }

    static ResampleHints parse_hints(const JVal* hints, const char* node) {
        ResampleHints r;
        if (!hints || hints->is_null()) return r;
        if (hints->t != JVal::Obj) raise(kInvalidJson, "InvalidJson: %s.hints must be an object", node);
        if (const JVal* s = hints->get("sharpen_percent")) if (s->t == JVal::Num) { r.has_sharpen = true; r.sharpen = static_cast<float>(s->n); }
        r.down = parse_filter(hints->get("down_filter"), r.down);
        r.up = parse_filter(hints->get("up_filter"), r.up);
        if (const JVal* cs = hints->get("scaling_colorspace"))
            if (cs->t == JVal::Str) {
                if (cs->s == "srgb") r.space = IFHIP_SPACE_SRGB;
                else if (cs->s != "linear") raise(kInvalidJson, "InvalidJson: scaling_colorspace must be srgb or linear");
            }
        if (const JVal* b = hints->get("background_color"))
            if (!b->is_null()) {
                r.has_bg = true;
                r.bg_keyword_transparent = b->t == JVal::Str && b->s == "transparent";
                r.bg = parse_color(b, "hints.background_color");
            }
        if (const JVal* rw = hints->get("resample_when"))
            if (rw->t == JVal::Str) {                                                // s::ResampleWhen (lib.rs:899-907)
                if (rw->s == "always") r.resample_when = ResampleHints::kAlways;
                else if (rw->s == "size_differs") r.resample_when = ResampleHints::kSizeDiffers;
                else if (rw->s == "size_differs_or_sharpening_requested") r.resample_when = ResampleHints::kSizeDiffersOrSharpen;
                else raise(kInvalidJson, "InvalidJson: unknown resample_when '%s'", rw->s.c_str());
            }
        if (const JVal* sw = hints->get("sharpen_when"))
            if (sw->t == JVal::Str) {                                                // s::SharpenWhen
                if (sw->s == "always") r.sharpen_when = ResampleHints::kSAlways;
                else if (sw->s == "downscaling") r.sharpen_when = ResampleHints::kSDown;
                else if (sw->s == "upscaling") r.sharpen_when = ResampleHints::kSUp;
                else if (sw->s == "size_differs") r.sharpen_when = ResampleHints::kSSizeDiffers;
                else raise(kInvalidJson, "InvalidJson: unknown sharpen_when '%s'", sw->s.c_str());
            }
        return r;
    }
    static float gated_sharpen(const ResampleHints& hi, uint32_t w, uint32_t h, uint32_t in_w, uint32_t in_h) {   // scale_render.rs:55-65, 263-274
        const bool size_differs = w != in_w || h != in_h, downscaling = w < in_w || h < in_h, upscaling = w > in_w || h > in_h;
        const float raw = hi.has_sharpen ? hi.sharpen : 0.f;
        switch (hi.sharpen_when) {
        case ResampleHints::kSAlways: return raw;
        case ResampleHints::kSDown: return downscaling ? raw : 0.f;
        case ResampleHints::kSUp: return upscaling ? raw : 0.f;
        case ResampleHints::kSSizeDiffers: return size_differs ? raw : 0.f;
        }
        return raw;
    }

    // DrawImageDef::render (flow/nodes/scale_render.rs:221-320): the one caller of the hot path.  compose = the
    // node's `blend` (None = Compose).
    void draw_image_exact(const FramePtr& canvas, const FramePtr& in, uint32_t x, uint32_t y, uint32_t w, uint32_t h, bool compose, const ResampleHints& hi) {
        Timed t(this, "draw_image_to_canvas");
        if (canvas == in) raise(kGraphInvalid, "InvalidNodeConnections: Canvas and Input are the same bitmap!");
        if (static_cast<uint64_t>(x) + w > canvas->w || static_cast<uint64_t>(y) + h > canvas->h)                       // :237-240
            raise(kArgumentInvalid, "InvalidNodeParams: DrawImageExact target rect x1=%u,y1=%u,w=%u,h=%u does not fit canvas size %ux%u.", x, y, w, h, canvas->w, canvas->h);
        if (w == 0 || h == 0) raise(kArgumentInvalid, "InvalidNodeParams: DrawImageExact target size must be non-zero");
        if (hi.resample_when != ResampleHints::kDefault && hi.resample_when != ResampleHints::kAlways)                 // :246-251
            raise(kArgumentInvalid, "InvalidNodeParams: DrawImageExact already has a canvas and cannot honor ResampleWhen");
        const bool upscaling = w > in->w || h > in->h;                                                                 // :253
        const int filter = upscaling ? hi.up : hi.down;                                                                // :257-261
        const float sharpen = gated_sharpen(hi, w, h, in->w, in->h);
        if (canvas->compose == IFHIP_REPLACE_SELF && compose) canvas->compose = IFHIP_BLEND_WITH_SELF;                 // :284-286
        if (canvas->compose == IFHIP_BLEND_WITH_MATTE && !compose && canvas->alpha) canvas->compose = IFHIP_REPLACE_SELF;   // :287-292
        poll_cancel();
        const std::shared_ptr<ifhip_resample_plan> plan_ref = shared_plan(in->w, in->h, w, h, filter, sharpen);
        ifhip_resample_plan* plan = plan_ref.get();
        if (in->pending) {                            // MzDec::read_frame + scale_and_render as one device call (mozjpeg_decoder.rs:346-362 -> :304-313)
            PendingJpeg& pj = *in->pending;
            int fused_call = 0;
            check(ifhip_jpeg_decode_resample_batch_device(pj.st, pj.coef[0], pj.coef[1], pj.coef[2], pj.d_qt, 1, plan, dev(canvas), canvas->bytes(), canvas->w,
                                                          canvas->h, canvas->stride, x, y, hi.space, canvas->compose, canvas->matte, &fused_call, t_job_stream));
            if (fused_call) c->fused_decode_resamples.fetch_add(1, std::memory_order_relaxed);
        } else
        check(ifhip_scale_and_render_batch_device(plan, in->d, in->bytes(), in->stride, in->alpha ? 1 : 0, 1, dev(canvas), canvas->bytes(),
                                                  canvas->w, canvas->h, canvas->stride, x, y, hi.space, canvas->compose, canvas->matte, nullptr, -1, t_job_stream));
        hip_check(static_cast<hipError_t>(ifhip::wait_stream(t_job_stream)), "draw_image_exact");
        canvas->compose = IFHIP_BLEND_WITH_SELF;                                                                       // :314
    }

    // Resample2D (scale_render.rs:30-120): removed from the graph unless it resamples or has a matte to apply to a
    // Bgra32 parent; otherwise CreateCanvas{parent.fmt, background_color} + Scale2d, which becomes DrawImageExact with
    // blend = Overwrite when the background is transparent (:139-201)
    FramePtr resample(const FramePtr& in, uint32_t w, uint32_t h, const JVal* hints_json) {
        if (w == 0 || h == 0) raise(kArgumentInvalid, "InvalidNodeParams: resample_2d target size must be non-zero");
        ResampleHints hi = parse_hints(hints_json, "resample_2d");
        const bool size_differs = w != in->w || h != in->h;
        const bool apply_matte = in->alpha && hi.has_bg && !hi.bg_keyword_transparent;                                 // :44-47
        const float sharpen = gated_sharpen(hi, w, h, in->w, in->h);
        const bool sharpen_requested = sharpen != 0.f;                                                                 // :67
        bool do_resample = false;
        switch (hi.resample_when) {                                                                                    // :69-78
        case ResampleHints::kAlways: do_resample = true; break;
        case ResampleHints::kSizeDiffers: do_resample = size_differs; break;
        default: do_resample = size_differs || sharpen_requested; break;
        }
        if (!do_resample && !apply_matte) return in;                                                                   // delete_node_and_snap_together (:115)
        hi.has_sharpen = true; hi.sharpen = sharpen;                                                                   // Some(sharpen_percent), sharpen_when passed on (:84-94)
        hi.resample_when = ResampleHints::kAlways;
        const uint32_t bg = hi.has_bg ? hi.bg : 0u;
        FramePtr canvas;
        {
            Timed t(this, "create_canvas");
            canvas = new_frame(w, h, in->alpha, bg, true, hi.has_bg && !hi.bg_keyword_transparent);                    // format: parent.fmt (:96-105)
        }
        draw_image_exact(canvas, in, 0, 0, w, h, (bg >> 24) != 0, hi);                                                 // blend Overwrite iff bgcolor.is_transparent() (:176-184)
        // the canvas keeps parent.fmt: nothing in scaling.rs:50-90 or DrawImageDef::render clears alpha_meaningful after an
        // opaque matte (only the encoder-side Bitmap::apply_matte does, bitmaps.rs:528-541), so a later node still sees Bgra32
        return canvas;
    }

    // constrain (flow/nodes/constrain.rs:41-98 -> imageflow_riapi process_constraint): the aspect-preserving modes that
    // need neither crop nor pad; sizes by AspectRatio::proportional / box_of (imageflow_riapi/src/sizing.rs:118-197).
    static void constrain_size(const std::string& m, uint32_t sw, uint32_t sh, bool has_w, bool has_h, int64_t tw, int64_t th, uint32_t* ow, uint32_t* oh) {
        int64_t w = sw, h = sh;
        if (m == "distort") { w = has_w ? tw : (has_h ? proportional(sw, sh, th, false, false, 0, 0) : sw); h = has_h ? th : (has_w ? proportional(sw, sh, tw, true, false, 0, 0) : sh); }
        else if (has_w && has_h) {
            if (m == "fit" || sw > tw || sh > th) inner_box(sw, sh, tw, th, &w, &h);         // within never up-scales
        } else if (has_w) {
            if (m == "fit" || sw > tw) { w = tw; h = proportional(sw, sh, tw, true, false, 0, 0); }
        } else if (has_h) {
            if (m == "fit" || sh > th) { h = th; w = proportional(sw, sh, th, false, false, 0, 0); }
        }
        *ow = static_cast<uint32_t>(w); *oh = static_cast<uint32_t>(h);
    }
    static void constrain_params(const JVal& p, const char* node, std::string* mode, bool* has_w, bool* has_h, int64_t* tw, int64_t* th) {
        const JVal* jm = p.get("mode");
        *mode = jm && jm->t == JVal::Str ? jm->s : "";
        const JVal *jw = p.get("w"), *jh = p.get("h");
        *has_w = jw && jw->t == JVal::Num; *has_h = jh && jh->t == JVal::Num;
        *tw = *has_w ? want_u32(p, "w", node) : 0; *th = *has_h ? want_u32(p, "h", node) : 0;
    }
    FramePtr constrain(const FramePtr& in, const JVal& p) {
        std::string m;
        bool has_w, has_h;
        int64_t tw, th;
        constrain_params(p, "constrain", &m, &has_w, &has_h, &tw, &th);
        const int mode = ifhip::constraint_mode_from_name(m);
        if (mode < 0) raise(kInvalidJson, "InvalidJson: unknown constrain mode '%s'", m.c_str());
        float gx = 50.f, gy = 50.f;
        const bool has_gravity = parse_gravity(p.get("gravity"), &gx, &gy);
        ifhip::ConstraintLayout lay;
        std::string err;
        if (!ifhip::process_constraint(mode, static_cast<int32_t>(in->w), static_cast<int32_t>(in->h), has_w ? tw : -1, has_h ? th : -1, has_gravity, gx, gy, &lay, &err))
            raise(kArgumentInvalid, "InvalidNodeParams: Constraint error: %s", err.c_str());                              // constrain.rs:50-52
        FramePtr f = in;
        if (lay.has_crop) {                                                                                                // :55-57
            Timed t(this, "crop_mutate");
            f = crop_frame(f, lay.crop[0], lay.crop[1], lay.crop[2], lay.crop[3]);
        }
        // canvas_color overrides the hints' background_color (:60-76)
        const JVal* hints = p.get("hints");
        const JVal* cc = p.get("canvas_color");
        const bool has_cc = cc && !cc->is_null();
        JVal merged;
        if (has_cc) {
            if (hints && hints->t == JVal::Obj) merged = *hints;
            merged.t = JVal::Obj;
            bool set = false;
            for (auto& kv : merged.o) if (kv.first == "background_color") { kv.second = *cc; set = true; }
            if (!set) merged.o.emplace_back("background_color", *cc);
            hints = &merged;
        }
        f = resample(f, static_cast<uint32_t>(lay.scale_w), static_cast<uint32_t>(lay.scale_h), hints);                    // :78-82
        if (lay.has_pad) {                                                                                                 // :84-92: canvas_color or Transparent
            Timed t(this, "expand_canvas");
            f = expand_frame(f, lay.pad[0], lay.pad[1], lay.pad[2], lay.pad[3], has_cc ? parse_color(cc, "constrain.canvas_color") : 0u, !has_cc || keyword_transparent(cc));
        }
        return f;
    }
    // ConstraintGravity (imageflow_types lib.rs:1054-1061): "center" | {"percentage": {x, y}} -> false for centre / absent
    static bool parse_gravity(const JVal* g, float* gx, float* gy) {
        if (!g || g->is_null() || (g->t == JVal::Str && g->s == "center")) return false;
        const JVal* pc = g->get("percentage");
        const JVal *jx = pc ? pc->get("x") : nullptr, *jy = pc ? pc->get("y") : nullptr;
        if (!jx || !jy || jx->t != JVal::Num || jy->t != JVal::Num) raise(kInvalidJson, "InvalidJson: gravity is \"center\" or {\"percentage\":{x,y}}");
        *gx = static_cast<float>(jx->n); *gy = static_cast<float>(jy->n);
        return true;
    }
    // Crop (clone_crop_fill_expand.rs:519-541), materialised as a copy that keeps the parent's state
    FramePtr crop_frame(const FramePtr& in, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {
        if (x2 <= x1 || y2 <= y1 || x2 > in->w || y2 > in->h) raise(kArgumentInvalid, "InvalidNodeParams: Invalid crop bounds");
        FramePtr cv = new_frame(x2 - x1, y2 - y1, in->alpha, 0, true);
        const int compose = in->compose;                                              // Bitmap::crop is a window onto the same bitmap
        const uint32_t matte = in->matte;                                             // (bitmaps.rs:841-859): its compositing mode stays
        copy_into_canvas(in, cv, x1, y1, x2 - x1, y2 - y1, 0, 0);
        cv->compose = compose; cv->matte = matte;
        return cv;
    }
    // ExpandCanvas (:224-262): CreateCanvas of the colour (Bgra32 unless the colour is opaque) + CopyRectToCanvas
    FramePtr expand_frame(const FramePtr& in, uint32_t l, uint32_t t2, uint32_t r, uint32_t b, uint32_t color, bool color_is_keyword_transparent) {
        const uint64_t nw = static_cast<uint64_t>(in->w) + l + r, nh = static_cast<uint64_t>(in->h) + t2 + b;
        check_size(sec.max_frame_size, "max_frame_size", nw, nh);                     // before the 32-bit sums can wrap
        FramePtr cv = new_frame(static_cast<uint32_t>(nw), static_cast<uint32_t>(nh), (color >> 24) == 255 ? in->alpha : true, color, true, !color_is_keyword_transparent);
        return copy_into_canvas(in, cv, 0, 0, in->w, in->h, l, t2);
    }

    // command_string {kind: "ir4", value: "width=200&..."}: the querystring form of BASELINE config 1.  Only the sizing
    // keys that reach the hot path (width/w, height/h, mode=max default; down.colorspace) -- imageflow_riapi is out of
    // scope.  JPEG pre-shrink hint exactly as Ir4Expand::get_decode_commands (imageflow_riapi/src/ir4/mod.rs:155-210).
    FramePtr command_string(const JVal& p, FramePtr in) {
        const JVal* kind = p.get("kind");
        const JVal* value = p.get("value");
        if (!kind || kind->t != JVal::Str || kind->s != "ir4" || !value || value->t != JVal::Str)
            raise(kInvalidJson, "InvalidJson: command_string needs kind \"ir4\" and a value");
        double qw = 0, qh = 0;
        bool srgb = false;
        std::string down_filter;                 // `down.filter` (ir4/parsing.rs:580 -> layout.rs:527 ResampleHints::down_filter)
        int quality = -1, jpeg_quality = -1;     // `quality` / `jpeg.quality` (ir4/encoder.rs:74: jpeg.quality, else quality)
        bool jpeg_out = false;                   // `format=jpg|jpeg`
        size_t i = 0;
        const std::string& q = value->s;
        while (i < q.size()) {
            const size_t amp = std::min(q.find('&', i), q.size());
            const std::string kv = q.substr(i, amp - i);
            i = amp + 1;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
            for (char& ch : k) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
            if (k == "width" || k == "w" || k == "maxwidth") qw = std::atof(v.c_str());
            else if (k == "height" || k == "h" || k == "maxheight") qh = std::atof(v.c_str());
            else if (k == "down.colorspace") srgb = v == "srgb";
            else if (k == "mode") { if (v != "max") raise(kActionNotSupported, "ActionNotSupported: querystring mode=%s (this shim: max)", v.c_str()); }
            else if (k == "down.filter") down_filter = querystring_filter_name(v);
            else if (k == "quality" || k == "jpeg.quality") {
                char* end = nullptr;
                const long q = std::strtol(v.c_str(), &end, 10);
                // a value that is no integer is ignored with a warning by the reference (ir4/parsing.rs parse_i32): the encoder's
                // default quality then applies -- the key still says "a JPEG comes out"
                if (end == v.c_str() || *end) jpeg_out = true;
                else (k == "quality" ? quality : jpeg_quality) = static_cast<int>(std::max(0l, std::min(100l, q)));
            }
            else if (k == "format") {
                for (char& ch : v) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
                if (v != "jpg" && v != "jpeg") raise(kActionNotSupported, "ActionNotSupported: querystring format=%s (this shim writes JPEG; PNG / GIF / WebP coders are out of scope)", v.c_str());
                jpeg_out = true;
            }
            else raise(kActionNotSupported, "ActionNotSupported: querystring key '%s'", k.c_str());
        }
        if (!(qw >= 0 && qw <= 2147483647.0) || !(qh >= 0 && qh <= 2147483647.0)) raise(kArgumentInvalid, "InvalidNodeParams: querystring width/height out of range");
        if (p.get("watermarks") && !p.get("watermarks")->is_null()) raise(kActionNotSupported, "ActionNotSupported: command_string.watermarks (use watermark nodes)");
        const JVal* dec = p.get("decode");
        const JVal* enc = p.get("encode");
        uint32_t src_w = 0, src_h = 0;
        // (the layout sees the frame BEHIND the decoder's orientation step, and the decoder hints are worked out from those
        // rotated sides and handed to the decoder as they are: command_string.rs:20-58, ir4/mod.rs:167-176)
        if (dec && dec->t == JVal::Num) image_size(static_cast<int32_t>(want_int(p, "decode", "command_string")), &src_w, &src_h, true);
        else if (in) { src_w = in->w; src_h = in->h; }
        else raise(kGraphInvalid, "GraphInvalid: command_string has neither a decode io nor an input frame");
        auto target = [&](uint32_t sw, uint32_t sh, uint32_t* ow, uint32_t* oh) {        // mode=max: fit inside, never up-scale
            constrain_size("within", sw, sh, qw >= 1, qh >= 1, static_cast<int64_t>(qw), static_cast<int64_t>(qh), ow, oh);
        };
        if (dec && dec->t == JVal::Num) {
            uint32_t hint_w = 0, hint_h = 0;
            if (src_w) {
                uint32_t ow, oh;
                target(src_w, src_h, &ow, &oh);
                const double downscale = std::min(static_cast<double>(src_w) / ow, static_cast<double>(src_h) / ow);   // sic: `to.w` twice (:161-162)
                const double preshrink = 2.1 / downscale;
                if (preshrink < 1.0) { hint_w = static_cast<uint32_t>(std::floor(src_w * preshrink)); hint_h = static_cast<uint32_t>(std::floor(src_h * preshrink)); }
            }
            in = decode_oriented(static_cast<int32_t>(dec->n), hint_w, hint_h, !srgb, !srgb);
            if (!src_w) { src_w = in->w; src_h = in->h; }
        }
        uint32_t ow, oh;
        target(src_w, src_h, &ow, &oh);
        JVal hints;
        hints.t = JVal::Obj;
        if (srgb) { JVal cs; cs.t = JVal::Str; cs.s = "srgb"; hints.o.emplace_back("scaling_colorspace", cs); }
        if (!down_filter.empty()) { JVal f; f.t = JVal::Str; f.s = down_filter; hints.o.emplace_back("down_filter", f); }
        {   // background_color: Some(bgcolor), Transparent unless the querystring names format=jpg, then white (ir4/layout.rs:492-503, :530)
            JVal bg;
            if (jpeg_out) { JVal hex; hex.t = JVal::Str; hex.s = "FFFFFFFF"; JVal srgb; srgb.t = JVal::Obj; srgb.o.emplace_back("hex", hex); bg.t = JVal::Obj; bg.o.emplace_back("srgb", srgb); }
            else { bg.t = JVal::Str; bg.s = "transparent"; }
            hints.o.emplace_back("background_color", bg);
        }
        FramePtr out = resample(in, ow, oh, &hints);
        if (enc && enc->t == JVal::Num) {
            // The reference keeps the source's format (a JPEG stays a JPEG, ir4/encoder.rs:30-37 OutputFormat::Keep) and hands
            // `jpeg.quality`, else `quality`, to its JPEG encoder (encoder.rs:74; 90 when neither is given, codecs/auto.rs).  Here a
            // querystring that names the format or a quality gets the classic JPEG writer with exactly that quality -- the
            // LIBJPEG-TURBO STYLE file (what the reference writes under jpeg.turbo=true); its default mozjpeg-style encoder
            // (trellis / scan search) is not built, so bytes and sizes differ from the reference's default for the same string
            // (README "Known differences", DESIGN "Encode").  One that names neither keeps this shim's labelled extension, the raw
            // BGRA container -- no key is accepted and then dropped.
            const int q = jpeg_quality >= 0 ? jpeg_quality : quality;
            if (jpeg_out || q >= 0) {
                JVal qv; qv.t = JVal::Num; qv.n = q >= 0 ? q : 90;
                JVal classic; classic.t = JVal::Obj; classic.o.emplace_back("quality", qv);
                JVal preset; preset.t = JVal::Obj; preset.o.emplace_back("libjpeg_turbo", classic);
                encode(out, static_cast<int32_t>(want_int(p, "encode", "command_string")), &preset, false);
            } else {
                encode(out, static_cast<int32_t>(want_int(p, "encode", "command_string")), nullptr, false);
            }
        }
        return out;
    }

    // encode: EncoderPreset::LibjpegTurbo is written as a real JPEG (codecs/mozjpeg.rs:78-160, create_classic :62-77):
    // apply_matte + forward DCT / quantisation on the device, the Huffman coder and the markers on the host
    // (csrc/jpeg_write.cpp) -- baseline, with optimize_huffman_coding libjpeg's optimal tables, with progressive its
    // standard scan script (:121-129).  The content-adaptive sampling choice
    // (evalchroma, an external crate) is not reproduced: the file uses the maximum the reference allows (:133, 4:2:0).
    // EXTENSION: every other preset writes the raw BGRA container (PNG / GIF / WebP coders are out of scope).
    // `shared`: other consumers still read this frame -- the matte is then applied to a private copy.
    void encode(FramePtr f, int32_t io_id, const JVal* preset, bool shared) {
        Timed t(this, "primitive_encoder");
        Io& o = output(io_id);
        if (o.out_state == OutState::Taken) raise(kArgumentInvalid, "InvalidArgument: Output buffer for io_id %d has already been taken", io_id);
        poll_cancel();
        const JVal* classic = preset ? preset->get("libjpeg_turbo") : nullptr;
        if (classic) {
            auto flag = [&](const char* k) { const JVal* v = classic->get(k); return v && v->t == JVal::Bool && v->b; };
            const int write_flags = (flag("progressive") ? IFHIP_JPEG_PROGRESSIVE : 0) | (flag("optimize_huffman_coding") ? IFHIP_JPEG_OPTIMIZE_HUFFMAN : 0);
            const JVal* q = classic->get("quality");
            int quality = 75;                                                            // mozjpeg.rs:32 DEFAULT_QUALITY
            // Option<i32> -> `q as u8` (codecs/auto.rs:201: the value WRAPS, 300 is 44 and -5 is 251) -> u8::min(100, ..) (mozjpeg.rs:71)
            if (q && q->t == JVal::Num) {
                if (q->n != std::floor(q->n) || q->n < -2147483648.0 || q->n > 2147483647.0) raise(kInvalidJson, "InvalidJson: encode.preset.libjpeg_turbo.quality is a 32-bit integer");
                quality = std::min(100, static_cast<int>(static_cast<uint8_t>(static_cast<int32_t>(q->n))));
            }
            const JVal* m = classic->get("matte");
            const uint32_t matte = m && !m->is_null() ? parse_color(m, "encode.preset.libjpeg_turbo.matte") : 0xFFFFFFFFu;   // :88-92
            if (shared && f->alpha) f = clone(f);
            check(ifhip_apply_matte_batch_device(dev(f), f->bytes(), 1, f->w, f->h, f->stride, f->alpha ? 1 : 0, matte, t_job_stream));
            f->alpha = false;                                                            // :94 set_alpha_meaningful(false)
            const uint8_t hs[3] = {2, 1, 1}, vs[3] = {2, 1, 1};
            uint16_t qt2[2][64], qt3[3][64];
            ifhip_jpeg_quality_tables(quality, &qt2[0][0]);
            std::memcpy(qt3[0], qt2[0], 128); std::memcpy(qt3[1], qt2[1], 128); std::memcpy(qt3[2], qt2[1], 128);
            ifhip_jpeg_fwd_stage* st = nullptr;
            check(ifhip_jpeg_fwd_stage_create(&st, f->w, f->h, hs, vs, 1));
            std::unique_ptr<ifhip_jpeg_fwd_stage, void (*)(ifhip_jpeg_fwd_stage*)> st_guard(st, ifhip_jpeg_fwd_stage_destroy);
            uint32_t bw[3], bh[3];
            check(ifhip_jpeg_fwd_stage_block_dims(st, bw, bh));
            size_t off[4] = {0, 0, 0, 0};
            for (int k = 0; k < 3; ++k) off[k + 1] = off[k] + static_cast<size_t>(bw[k]) * bh[k] * 64u;
            int16_t* d_coef = nullptr;
            uint16_t* d_qt = nullptr;
            hip_check(job_malloc(reinterpret_cast<void**>(&d_coef), off[3] * 2u + 384u), "hipMalloc(coefficients)");
            std::unique_ptr<int16_t, void (*)(int16_t*)> coef_guard(d_coef, [](int16_t* p) { job_free(p); });
            d_qt = reinterpret_cast<uint16_t*>(d_coef + off[3]);
            hip_check(static_cast<hipError_t>(ifhip::copy_to_device(d_qt, qt3, 384)), "upload(quant tables)");
            check(ifhip_jpeg_forward_batch_device(st, dev(f), f->bytes(), f->stride, d_qt, 1, d_coef + off[0], d_coef + off[1], d_coef + off[2], t_job_stream));
            poll_cancel();
            if (write_flags == 0) {
                // the preset's default (baseline, Annex K tables): the device entropy coder -- only the file leaves the device.
                // First with room for a scan half the size of its coefficients (what the host path assumes too), then, for an
                // image that is denser than that, with the geometry's worst case.
                for (int attempt = 0; attempt < 2; ++attempt) {
                    const size_t scan_cap = attempt == 0 ? std::max<size_t>(65536u, off[3]) : 0u;
                    // A geometry the coder does not take (more than 2.58 M blocks: a `security` override above ~110 MP) or a
                    // stage that cannot be allocated is not the job's failure: the host writer below codes the same file.
                    ifhip_jpeg_enc_stage* es = nullptr;
                    if (ifhip_jpeg_enc_stage_create(&es, f->w, f->h, 3, hs, vs, bw, bh, 1, scan_cap) != IFHIP_OK) break;
                    std::unique_ptr<ifhip_jpeg_enc_stage, void (*)(ifhip_jpeg_enc_stage*)> es_guard(es, [](ifhip_jpeg_enc_stage* q) { quiesce(); ifhip_jpeg_enc_stage_destroy(q); });
                    const size_t pitch = ifhip_jpeg_enc_stage_max_file_bytes(es);
                    uint8_t* d_file = nullptr;
                    if (job_malloc(reinterpret_cast<void**>(&d_file), pitch + 16u) != hipSuccess) break;
                    std::unique_ptr<uint8_t, void (*)(uint8_t*)> file_guard(d_file, [](uint8_t* p) { job_free(p); });
                    uint32_t* d_len = reinterpret_cast<uint32_t*>(d_file + ((pitch + 3u) & ~static_cast<size_t>(3u)));   // length, status behind the file
                    check(ifhip_jpeg_encode_batch_device(es, d_coef + off[0], d_coef + off[1], d_coef + off[2], quality, 1, d_file, pitch, d_len, d_len + 1, t_job_stream));
                    uint32_t len_status[2] = {0, 0};
                    hip_check(static_cast<hipError_t>(ifhip::copy_to_host(len_status, d_len, 8)), "download(file length)");
                    if (len_status[1] & IFHIP_ENC_BAD_COEFFICIENT)
                        raise(kArgumentInvalid, "InvalidArgument: coefficient out of range for 8-bit JPEG (more than 11 DC / 10 AC magnitude bits)");
                    if (len_status[1] != 0) continue;
                    o.owned.assign(len_status[0], 0);
                    hip_check(static_cast<hipError_t>(ifhip::copy_to_host(o.owned.data(), d_file, len_status[0])), "download(file)");
                    o.written = true;
                    encodes.push_back({io_id, f->w, f->h, "image/jpeg", "jpg"});
                    c->device_coded_files.fetch_add(1, std::memory_order_relaxed);
                    return;
                }
            }                                                                            // (not coded on the device: the host writer)
            std::vector<int16_t> coef(off[3]);
            hip_check(static_cast<hipError_t>(ifhip::copy_to_host(coef.data(), d_coef, off[3] * 2u)), "download(coefficients)");
            size_t len = 0;
            o.owned.assign(std::max<size_t>(4096u, off[3]), 0);                          // a file is smaller than its coefficients: one pass
            int wrc = ifhip_jpeg_write(coef.data() + off[0], coef.data() + off[1], coef.data() + off[2], bw, bh, 3, hs, vs, f->w, f->h, quality, write_flags,
                                       o.owned.data(), o.owned.size(), &len);
            if (wrc != IFHIP_OK && len > o.owned.size()) {
                o.owned.assign(len, 0);
                wrc = ifhip_jpeg_write(coef.data() + off[0], coef.data() + off[1], coef.data() + off[2], bw, bh, 3, hs, vs, f->w, f->h, quality, write_flags,
                                       o.owned.data(), o.owned.size(), &len);
            }
            check(wrc);
            o.owned.resize(len);
            o.written = true;
            encodes.push_back({io_id, f->w, f->h, "image/jpeg", "jpg"});
            return;
        }
        o.owned.assign(kRawHeader + f->bytes(), 0);
        std::memcpy(o.owned.data(), kRawMagic, 8);
        const uint32_t hdr[4] = {f->w, f->h, f->stride, f->alpha ? 1u : 0u};
        std::memcpy(o.owned.data() + 8, hdr, 16);
        { uint8_t* src = dev(f); hip_check(static_cast<hipError_t>(ifhip::copy_to_host(o.owned.data() + kRawHeader, src, f->bytes())), "download(frame)"); }
        o.written = true;
        encodes.push_back({io_id, f->w, f->h, "application/x-imageflow-bgra", "ifbgra"});
    }

    // CopyRectNodeDef::render (flow/nodes/clone_crop_fill_expand.rs:31-90)
    FramePtr copy_into_canvas(const FramePtr& in, const FramePtr& canvas, uint32_t fx, uint32_t fy, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
        if (in == canvas) raise(kGraphInvalid, "InvalidNodeConnections: Canvas and Input are the same bitmap!");
        if (in->w <= fx || in->h <= fy || in->w < static_cast<uint64_t>(fx) + w || in->h < static_cast<uint64_t>(fy) + h ||
            canvas->w < static_cast<uint64_t>(x) + w || canvas->h < static_cast<uint64_t>(y) + h)
            raise(kArgumentInvalid, "InvalidNodeParams: Invalid coordinates. Canvas is %ux%u, Input is %ux%u, Params provided: from_x=%u from_y=%u w=%u h=%u x=%u y=%u",
                  canvas->w, canvas->h, in->w, in->h, fx, fy, w, h, x, y);
        int canvas_alpha = canvas->alpha ? 1 : 0;
        check(ifhip_copy_rect_batch_device(dev(in), in->bytes(), in->w, in->h, in->stride, in->alpha ? 1 : 0, dev(canvas), canvas->bytes(),
                                           canvas->w, canvas->h, canvas->stride, &canvas_alpha, fx, fy, x, y, w, h, 1, t_job_stream));
        canvas->alpha = canvas_alpha != 0;
        canvas->compose = IFHIP_BLEND_WITH_SELF;                                       // copy_rect.rs:37
        hip_check(static_cast<hipError_t>(ifhip::wait_stream(t_job_stream)), "copy_rect");
        return canvas;
    }
    FramePtr transposed(const FramePtr& in) {
        FramePtr t = new_frame(in->h, in->w, in->alpha, 0, true);
        check(ifhip_transpose_batch_device(dev(in), in->bytes(), in->w, in->h, in->stride, t->d, t->bytes(), t->w, t->h, t->stride, 1, t_job_stream));
        return t;
    }
    FramePtr flip(const FramePtr& in, bool vertical) {
        check(vertical ? ifhip_flip_vertical_batch_device(dev(in), in->bytes(), 1, in->w, in->h, in->stride, t_job_stream)
                       : ifhip_flip_horizontal_batch_device(dev(in), in->bytes(), 1, in->w, in->h, in->stride, t_job_stream));
        return in;
    }
    // ApplyOrientationDef::expand (flow/nodes/rotate_flip_transpose.rs:44-66): the EXIF flag as flips and a transpose
    FramePtr apply_orientation(const FramePtr& in, int64_t flag) {
        switch (flag) {
        case 2: return flip(in, false);
        case 3: return flip(flip(in, true), false);
        case 4: return flip(in, true);
        case 5: return transposed(in);
        case 6: return flip(transposed(in), false);
        case 7: return transposed(flip(flip(in, true), false));
        case 8: return flip(transposed(in), true);
        default: return in;
        }
    }
    // DecoderDef::expand (flow/nodes/codecs_and_pointer.rs:94-107): a decode whose input carries an EXIF orientation
    // above 0 is followed by ApplyOrientation {flag}.  Flag 1 is the identity -- the frame stays a pending JPEG, so that
    // decode + resample remain one device call; the others need the bitmap.
    FramePtr decode_oriented(int32_t io_id, uint32_t hint_w, uint32_t hint_h, bool luma_spatial, bool luma_srgb) {
        FramePtr f = decode(io_id, hint_w, hint_h, luma_spatial, luma_srgb);
        const int flag = exif_flag(input(io_id));
        if (flag < 2) return f;
        Timed t(this, "apply_orientation");
        return apply_orientation(f, flag);
    }
    // ColorMatrixSrgbMutDef::mutate (flow/nodes/color.rs:20-38)
    FramePtr color_matrix(const FramePtr& in, const float m[25]) {
        check(ifhip_apply_color_matrix_batch_device(dev(in), in->bytes(), 1, in->w, in->h, in->stride, m, t_job_stream));
        in->compose = IFHIP_BLEND_WITH_SELF;
        return in;
    }
    // ColorFilterSrgb::expand (flow/nodes/color.rs:49-83) with the matrices of :86-230
    FramePtr color_filter(const FramePtr& in, const JVal& p) {
        float m[25] = {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
        auto gray = [&](float r, float g, float b) { const float v[25] = {r, r, r, 0, 0, g, g, g, 0, 0, b, b, b, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1}; std::memcpy(m, v, sizeof v); };
        std::string name;
        float a = 0.f;
        if (p.t == JVal::Str) name = p.s;
        else if (p.t == JVal::Obj && p.o.size() == 1 && p.o[0].second.t == JVal::Num) { name = p.o[0].first; a = static_cast<float>(p.o[0].second.n); }
        else raise(kInvalidJson, "InvalidJson: color_filter_srgb is a name or {name: value}");
        if (name == "sepia") { const float v[25] = {0.393f, 0.349f, 0.272f, 0, 0, 0.769f, 0.686f, 0.534f, 0, 0, 0.189f, 0.168f, 0.131f, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0}; std::memcpy(m, v, sizeof v); }
        else if (name == "grayscale_ntsc") gray(0.229f, 0.587f, 0.114f);
        else if (name == "grayscale_ry") gray(0.5f, 0.419f, 0.081f);
        else if (name == "grayscale_flat") gray(0.5f, 0.5f, 0.5f);
        else if (name == "grayscale_bt709") gray(0.2125f, 0.7154f, 0.0721f);
        else if (name == "invert") { m[0] = m[6] = m[12] = -1.f; m[20] = m[21] = m[22] = 1.f; }
        else if (name == "alpha") m[18] = a;
        else if (name == "contrast") { const float c2 = a + 1.f, t = 0.5f * (1.f - c2); m[0] = m[6] = m[12] = c2; m[20] = m[21] = m[22] = t; }
        else if (name == "brightness") m[20] = m[21] = m[22] = a;
        else if (name == "saturation") {
            const float s = std::max(a + 1.f, 0.f), c2 = 1.f - s, cr = 0.3086f * c2, cg = 0.6094f * c2, cb = 0.0820f * c2;
            const float v[25] = {cr + s, cr, cr, 0, 0, cg, cg + s, cg, 0, 0, cb, cb, cb + s, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
            std::memcpy(m, v, sizeof v);
        } else raise(kInvalidJson, "InvalidJson: unknown color_filter_srgb '%s'", name.c_str());
        if (name == "alpha" && !in->alpha) {                                          // EnableTransparency (enable_transparency.rs:68-95)
            check(ifhip_normalize_unused_alpha_batch_device(dev(in), in->bytes(), 1, in->w, in->h, in->stride, 0, t_job_stream));
            in->alpha = true;
        }
        return color_matrix(in, m);
    }

    // WatermarkDef::expand (flow/nodes/watermark.rs:100-196): decode -> [crop] -> [alpha(opacity)] -> DrawImageExact
    // (Compose) onto the input frame.  As the reference, the constraint is built with `hints: None`, so the node's
    // own `hints` never reach the resampler (:121-128, :176).
    FramePtr watermark(const FramePtr& canvas, const JVal& p) {
        const int32_t io_id = static_cast<int32_t>(want_int(p, "io_id", "watermark"));
        auto opt_u32 = [&](const char* k) -> uint32_t { const JVal* v = p.get(k); return v && v->t == JVal::Num ? want_u32(p, k, "watermark") : 0u; };
        if (!(opt_u32("min_canvas_width") < canvas->w && opt_u32("min_canvas_height") < canvas->h)) return canvas;       // :109-111
        // get_bounding_box (:11-58)
        int64_t bx1 = 0, by1 = 0, bx2 = canvas->w, by2 = canvas->h;
        if (const JVal* fb = p.get("fit_box"))
            if (!fb->is_null()) {
                const JVal* margins = fb->get("image_margins") ? fb->get("image_margins") : fb->get("canvas_margins");
                const JVal* pct = fb->get("image_percentage") ? fb->get("image_percentage") : fb->get("canvas_percentage");
                if (margins) {
                    const int64_t l = want_u32(*margins, "left", "fit_box"), t = want_u32(*margins, "top", "fit_box"), r = want_u32(*margins, "right", "fit_box"), b = want_u32(*margins, "bottom", "fit_box");
                    if (!(l + r < canvas->w && t + b < canvas->h)) return canvas;
                    bx1 = l; by1 = t; bx2 = static_cast<int64_t>(canvas->w) - r; by2 = static_cast<int64_t>(canvas->h) - b;
                } else if (pct) {
                    auto num = [&](const char* k) { const JVal* v = pct->get(k); if (!v || v->t != JVal::Num) raise(kInvalidJson, "InvalidJson: fit_box.%s must be a number", k); return static_cast<float>(v->n); };
                    auto to_px = [](float percent, uint32_t size) { const float ratio = std::min(std::max(percent, 0.f), 100.f) / 100.f; return static_cast<int64_t>(std::round(ratio * static_cast<float>(size))); };
                    bx1 = to_px(num("x1"), canvas->w); by1 = to_px(num("y1"), canvas->h); bx2 = to_px(num("x2"), canvas->w); by2 = to_px(num("y2"), canvas->h);
                    if (!(bx1 < bx2 && by1 < by2)) return canvas;
                } else raise(kInvalidJson, "InvalidJson: unknown watermark fit_box");
            }
        const JVal* fm = p.get("fit_mode");
        const std::string mode = fm && fm->t == JVal::Str ? fm->s : "within";                                            // :121
        if (mode != "within" && mode != "fit" && mode != "distort" && mode != "within_crop" && mode != "fit_crop")       // WatermarkConstraintMode (lib.rs:1022-1040)
            raise(kInvalidJson, "InvalidJson: unknown watermark fit_mode '%s'", mode.c_str());
        uint32_t mw = 0, mh = 0;
        image_size(io_id, &mw, &mh, true);                                                                                // get_scaled_rotated_image_info (:128)
        float gx = 50.f, gy = 50.f;                                                                                       // obey_gravity (:69-86)
        const bool has_gravity = parse_gravity(p.get("gravity"), &gx, &gy);
        // Constraint {mode, w: box_w, h: box_h, hints: None, gravity, canvas_color: None} -> process_constraint (:121-139)
        ifhip::ConstraintLayout lay;
        std::string err;
        if (!ifhip::process_constraint(ifhip::constraint_mode_from_name(mode), static_cast<int32_t>(mw), static_cast<int32_t>(mh), bx2 - bx1, by2 - by1,
                                       has_gravity, gx, gy, &lay, &err))
            raise(kArgumentInvalid, "InvalidNodeParams: Constraint error: %s", err.c_str());                              // (the reference unwraps: a panic)
        const uint32_t w = static_cast<uint32_t>(lay.scale_w), h = static_cast<uint32_t>(lay.scale_h);
        auto gravity1d = [](float pct, int64_t inner, int64_t outer) -> int64_t {                                         // :60-67
            const float ratio = std::min(std::max(pct, 0.f), 100.f) / 100.f;
            if ((outer < inner && inner < 1) || outer < 1) raise(kArgumentInvalid, "InvalidNodeParams: Watermark fit_box does not work");
            return static_cast<int64_t>(std::round(static_cast<float>(outer - inner) * ratio));
        };
        const int64_t x1 = gravity1d(gx, w, bx2 - bx1) + bx1, y1 = gravity1d(gy, h, by2 - by1) + by1;
        if (x1 < 0 || y1 < 0) raise(kArgumentInvalid, "InvalidNodeParams: Watermark fit_box does not work");
        FramePtr mark = decode_oriented(io_id, 0, 0, false, false);
        if (lay.has_crop) {                                                                                              // :157-164
            Timed t(this, "crop_mutate");
            mark = crop_frame(mark, lay.crop[0], lay.crop[1], lay.crop[2], lay.crop[3]);
        }
        float opacity = 1.f;
        if (const JVal* o = p.get("opacity")) if (o->t == JVal::Num) opacity = std::min(std::max(static_cast<float>(o->n), 0.f), 1.f);
        if (opacity < 1.f) {                                                                                             // :166-172
            Timed t(this, "color_matrix_srgb_mut");
            JVal f; f.t = JVal::Obj;
            JVal v; v.t = JVal::Num; v.n = static_cast<double>(opacity);
            f.o.emplace_back("alpha", v);
            color_filter(mark, f);
        }
        draw_image_exact(canvas, mark, static_cast<uint32_t>(x1), static_cast<uint32_t>(y1), w, h, true, ResampleHints{});
        return canvas;
    }

    // one node: `in` is the frame of its input edge (null for source nodes), `canvas` the frame of its canvas edge
    FramePtr run_node(const std::string& name, const JVal& p, FramePtr in, FramePtr canvas, bool in_shared) {
        auto need_input = [&] { if (!in) raise(kGraphInvalid, "GraphInvalid: node '%s' has no input frame", name.c_str()); };
        auto need_canvas = [&] { if (!canvas) raise(kGraphInvalid, "InvalidNodeConnections: node '%s' needs a canvas edge", name.c_str()); };
        poll_cancel();                                                                // execution_engine.rs:502
        if (name == "draw_image_exact") {                                             // s::Node::DrawImageExact (lib.rs:1328-1336)
            need_input(); need_canvas();
            const JVal* blend = p.get("blend");
            bool compose = true;
            if (blend && blend->t == JVal::Str) { if (blend->s == "overwrite") compose = false; else if (blend->s != "compose") raise(kInvalidJson, "InvalidJson: blend is compose or overwrite"); }
            draw_image_exact(canvas, in, want_u32(p, "x", "draw_image_exact"), want_u32(p, "y", "draw_image_exact"), want_u32(p, "w", "draw_image_exact"),
                             want_u32(p, "h", "draw_image_exact"), compose, parse_hints(p.get("hints"), "draw_image_exact"));
            return canvas;
        }
        if (name == "copy_rect_to_canvas") {
            need_input(); need_canvas();
            Timed t(this, "copy_rect_to_canvas");
            return copy_into_canvas(in, canvas, want_u32(p, "from_x", name.c_str()), want_u32(p, "from_y", name.c_str()), want_u32(p, "w", name.c_str()),
                                    want_u32(p, "h", name.c_str()), want_u32(p, "x", name.c_str()), want_u32(p, "y", name.c_str()));
        }
        if (canvas) raise(kGraphInvalid, "InvalidNodeConnections: node '%s' does not take a canvas edge", name.c_str());
        if (name == "decode") {
            uint32_t hw = 0, hh = 0;
            bool spatial = false, gamma = false;
            {
                const Io& told = input(static_cast<int32_t>(want_int(p, "io_id", "decode")));
                if (told.told) { hw = told.told_w; hh = told.told_h; spatial = told.told_spatial; gamma = told.told_gamma; }
            }
            if (const JVal* cmds = p.get("commands")) {
                if (cmds->t == JVal::Arr) {
                    for (const JVal& cmd : cmds->a) {
                        if (cmd.t == JVal::Str && cmd.s == "discard_color_profile") {
                            input(static_cast<int32_t>(want_int(p, "io_id", "decode"))).told_discard_profile = true;
                        } else if (const JVal* j = cmd.get("jpeg_downscale_hints")) {        // s::JpegIDCTDownscaleHints
                            hw = want_u32(*j, "width", "jpeg_downscale_hints"); hh = want_u32(*j, "height", "jpeg_downscale_hints");
                            if (const JVal* b = j->get("scale_luma_spatially")) spatial = b->t == JVal::Bool && b->b;
                            if (const JVal* b = j->get("gamma_correct_for_srgb_during_spatial_luma_scaling")) gamma = b->t == JVal::Bool && b->b;
                        }
                    }
                }
            }
            return decode_oriented(static_cast<int32_t>(want_int(p, "io_id", "decode")), hw, hh, spatial, gamma);
        }
        if (name == "create_canvas") {
            Timed t(this, "create_canvas");
            const JVal* fmt = p.get("format");
            const std::string f = fmt && fmt->t == JVal::Str ? fmt->s : "bgra_32";
            if (f != "bgra_32" && f != "bgr_32") raise(kActionNotSupported, "ActionNotSupported: create_canvas format %s", f.c_str());
            return new_frame(want_u32(p, "w", "create_canvas"), want_u32(p, "h", "create_canvas"), f == "bgra_32", parse_color(p.get("color"), "create_canvas.color"), true,
                             !keyword_transparent(p.get("color")));
        }
        if (name == "command_string") return command_string(p, in);
        need_input();
        if (name == "resample_2d") return resample(in, want_u32(p, "w", "resample_2d"), want_u32(p, "h", "resample_2d"), p.get("hints"));
        if (name == "constrain") return constrain(in, p);
        if (name == "watermark") return watermark(in, p);
        if (name == "encode") { encode(in, static_cast<int32_t>(want_int(p, "io_id", "encode")), p.get("preset"), in_shared); return in; }
        Timed t(this, name == "fill_rect" ? "fill_rect_mutate" : name == "crop" ? "crop_mutate" : name == "flip_v" ? "flip_vertical_mutate" : name == "flip_h" ? "flip_vertical_mutate" /* sic: rotate_flip_transpose.rs:206 */
                      : name == "color_matrix_srgb" || name == "color_filter_srgb" ? "color_matrix_srgb_mut" : name == "expand_canvas" || name == "region" || name == "region_percent" ? "expand_canvas" : name == "transpose" ? "transpose_mut"
                      : name == "rotate_90" ? "rotate_90" : name == "rotate_180" ? "rotate_180" : name == "rotate_270" ? "rotate_270" : name == "apply_orientation" ? "apply_orientation" : "node");
        if (name == "fill_rect") {                                                    // clone_crop_fill_expand.rs:107-137
            in->compose = IFHIP_BLEND_WITH_SELF;                                      // :112: set before the fill, so matte canvases accept sub-rects
            {   // the node's own check (:114-127): an empty rectangle is an error HERE (fill_rectangle itself lets one pass)
                const uint32_t x1 = want_u32(p, "x1", name.c_str()), y1 = want_u32(p, "y1", name.c_str()), x2 = want_u32(p, "x2", name.c_str()), y2 = want_u32(p, "y2", name.c_str());
                if (x2 <= x1 || y2 <= y1 || static_cast<int32_t>(x1) < 0 || static_cast<int32_t>(y1) < 0 || x2 > in->w || y2 > in->h)
                    raise(kArgumentInvalid, "InvalidCoordinates: Invalid coordinates for %ux%u bitmap: fill_rect x1=%u y1=%u x2=%u y2=%u", in->w, in->h, x1, y1, x2, y2);
            }
            check(ifhip_fill_rect_batch_device(dev(in), in->bytes(), 1, in->w, in->h, in->stride, in->compose, want_u32(p, "x1", name.c_str()),
                                               want_u32(p, "y1", name.c_str()), want_u32(p, "x2", name.c_str()), want_u32(p, "y2", name.c_str()),
                                               parse_color(p.get("color"), "fill_rect.color"), t_job_stream));
            return in;
        }
        if (name == "expand_canvas") {                                                // :224-262
            const uint32_t l = want_u32(p, "left", "expand_canvas"), t2 = want_u32(p, "top", "expand_canvas"), r = want_u32(p, "right", "expand_canvas"),
                           b = want_u32(p, "bottom", "expand_canvas"), color = parse_color(p.get("color"), "expand_canvas.color");
            return expand_frame(in, l, t2, r, b, color, keyword_transparent(p.get("color")));
        }
        if (name == "crop") {                                                         // :519-541 (materialised: a copy)
            const uint32_t x1 = want_u32(p, "x1", "crop"), y1 = want_u32(p, "y1", "crop"), x2 = want_u32(p, "x2", "crop"), y2 = want_u32(p, "y2", "crop");
            FramePtr cv = crop_frame(in, x1, y1, x2, y2);
            if (in_shared) { cv->compose = IFHIP_BLEND_WITH_SELF; cv->matte = 0; }    // CROP is MutProtect (clone_crop_fill_expand.rs:6): a window onto the Clone
            return cv;
        }
        if (name == "region" || name == "region_percent") {                            // :263-452
            // RegionPercent rewrites itself into Region with pixel corners (get_coords :265-286: f32 arithmetic, round half
            // away from zero, a side the percentages collapse gets one pixel), Region into Crop + ExpandCanvas -- or into a
            // plain CreateCanvas of the parent's format when the rectangle misses the frame altogether (:409-421)
            int64_t x1, y1, x2, y2;
            const uint32_t color = parse_color(p.get("background_color"), "region.background_color");
            if (name == "region_percent") {
                auto pct = [&](const char* k) -> float {
                    const JVal* v = p.get(k);
                    if (!v || v->t != JVal::Num) raise(kInvalidJson, "InvalidJson: region_percent.%s is a number", k);
                    return static_cast<float>(v->n);
                };
                const float l = pct("x1"), t2 = pct("y1"), r = pct("x2"), b = pct("y2");
                if (b <= t2 || r <= l) raise(kArgumentInvalid, "InvalidNodeParams: Invalid coordinates: %g,%g %g,%g should describe the top-left and bottom-right corners of the region in percentages. Not a rectangle.", l, t2, r, b);
                auto px = [](uint32_t side, float pc) -> int64_t {                    // `(side as f32 * pc / 100f32).round() as i32` (saturating)
                    const float v = std::round(static_cast<float>(side) * pc / 100.0f);
                    return v != v ? 0 : v >= 2147483648.0f ? INT32_MAX : v <= -2147483648.0f ? INT32_MIN : static_cast<int64_t>(v);
                };
                x1 = px(in->w, l); y1 = px(in->h, t2); x2 = px(in->w, r); y2 = px(in->h, b);
                if (x2 < x1) x2 = x1 + 1;                                             // sic: `<`, equal corners fall through to Region's own check
                if (y2 < y1) y2 = y1 + 1;
            } else {
                auto i32 = [&](const char* k) -> int64_t {
                    const int64_t v = want_int(p, k, "region");
                    if (v < INT32_MIN || v > INT32_MAX) raise(kInvalidJson, "InvalidJson: region.%s is a 32-bit integer", k);
                    return v;
                };
                x1 = i32("x1"); y1 = i32("y1"); x2 = i32("x2"); y2 = i32("y2");
            }
            if (y2 <= y1 || x2 <= x1) raise(kArgumentInvalid, "InvalidNodeParams: Invalid coordinates: %lld,%lld %lld,%lld should describe the top-left and bottom-right corners of the region in pixels. Not a rectangle.",
                                            static_cast<long long>(x1), static_cast<long long>(y1), static_cast<long long>(x2), static_cast<long long>(y2));
            const int64_t iw = in->w, ih = in->h;
            check_size(sec.max_frame_size, "max_frame_size", static_cast<uint64_t>(x2 - x1), static_cast<uint64_t>(y2 - y1));
            if (x1 >= iw || y1 >= ih || x2 <= 0 || y2 <= 0)                           // nothing of the input inside: a canvas of the colour
                return new_frame(static_cast<uint32_t>(x2 - x1), static_cast<uint32_t>(y2 - y1), in->alpha, color, true, !keyword_transparent(p.get("background_color")));
            const uint32_t cx1 = static_cast<uint32_t>(std::min(iw, std::max<int64_t>(0, x1))), cy1 = static_cast<uint32_t>(std::min(ih, std::max<int64_t>(0, y1)));
            const uint32_t cx2 = static_cast<uint32_t>(std::min(iw, std::max<int64_t>(0, x2))), cy2 = static_cast<uint32_t>(std::min(ih, std::max<int64_t>(0, y2)));
            const uint32_t el = static_cast<uint32_t>(std::max<int64_t>(0, -x1)), et = static_cast<uint32_t>(std::max<int64_t>(0, -y1));
            const uint32_t er = static_cast<uint32_t>(std::max<int64_t>(0, x2 - iw)), eb = static_cast<uint32_t>(std::max<int64_t>(0, y2 - ih));
            FramePtr part = in;
            if (in_shared || cx1 != 0 || cy1 != 0 || cx2 != in->w || cy2 != in->h) {  // Crop (a full-frame crop is the frame -- unless the frame has other readers: CROP is MutProtect, and copy_rectangle below normalises its input's unused alpha in place)
                part = new_frame(cx2 - cx1, cy2 - cy1, in->alpha, 0, true);
                copy_into_canvas(in, part, cx1, cy1, cx2 - cx1, cy2 - cy1, 0, 0);
            }
            // ExpandCanvas, also by nothing: CreateCanvas of the colour + CopyRectToCanvas (:231-255)
            FramePtr cv = new_frame(part->w + el + er, part->h + et + eb, (color >> 24) == 255 ? part->alpha : true, color, true, !keyword_transparent(p.get("background_color")));
            return copy_into_canvas(part, cv, 0, 0, part->w, part->h, el, et);
        }
        if (name == "color_matrix_srgb") {                                            // s::Node::ColorMatrixSrgb {matrix: [[f32;5];5]}
            const JVal* mj = p.get("matrix");
            float m[25];
            if (!mj || mj->t != JVal::Arr || mj->a.size() != 5) raise(kInvalidJson, "InvalidJson: color_matrix_srgb.matrix is 5 rows of 5 numbers");
            for (int r = 0; r < 5; ++r) {
                const JVal& row = mj->a[static_cast<size_t>(r)];
                if (row.t != JVal::Arr || row.a.size() != 5) raise(kInvalidJson, "InvalidJson: color_matrix_srgb.matrix is 5 rows of 5 numbers");
                for (int k = 0; k < 5; ++k) { if (row.a[static_cast<size_t>(k)].t != JVal::Num) raise(kInvalidJson, "InvalidJson: color_matrix_srgb.matrix is 5 rows of 5 numbers"); m[r * 5 + k] = static_cast<float>(row.a[static_cast<size_t>(k)].n); }
            }
            return color_matrix(in, m);
        }
        if (name == "color_filter_srgb") return color_filter(in, p);
        if (name == "flip_v") return flip(in, true);
        if (name == "flip_h") return flip(in, false);
        if (name == "transpose") return transposed(in);
        if (name == "rotate_90") return flip(transposed(in), false);                  // rotate_flip_transpose.rs:51-66
        if (name == "rotate_180") return flip(flip(in, true), false);
        if (name == "rotate_270") return flip(transposed(in), true);
        if (name == "apply_orientation") return apply_orientation(in, want_int(p, "flag", "apply_orientation"));
        raise(kActionNotSupported, "ActionNotSupported: node '%s' is outside the pixel hot path this library replaces", name.c_str());
    }

    static void node_of(const JVal& n, std::string* name, const JVal** params) {
        if (n.t == JVal::Str) { *name = n.s; static const JVal kNull; *params = &kNull; return; }
        if (n.t != JVal::Obj || n.o.size() != 1) raise(kInvalidJson, "InvalidJson: a node is {\"name\": {params}}");
        *name = n.o[0].first;
        *params = &n.o[0].second;
    }
    static bool mutates_input(const std::string& nm) {
        return nm == "fill_rect" || nm == "flip_v" || nm == "flip_h" || nm == "rotate_180" || nm == "color_matrix_srgb" || nm == "color_filter_srgb" ||
               nm == "apply_orientation" || nm == "watermark";
    }

    void run_framewise(const JVal& fw) {
        if (const JVal* steps = fw.get("steps")) {
            if (steps->t != JVal::Arr) raise(kInvalidJson, "InvalidJson: framewise.steps must be an array");
            FramePtr cur;
            for (const JVal& n : steps->a) {
                std::string name;
                const JVal* params;
                node_of(n, &name, &params);
                lazy_decode = true;                                                    // a chain: every frame has one consumer
                cur = run_node(name, *params, cur, nullptr, false);
            }
            return;
        }
        const JVal* graph = fw.get("graph");
        if (!graph) raise(kInvalidJson, "InvalidJson: framewise needs steps or graph");
        const JVal *nodes = graph->get("nodes"), *edges = graph->get("edges");
        if (!nodes || nodes->t != JVal::Obj || !edges || edges->t != JVal::Arr) raise(kInvalidJson, "InvalidJson: graph needs nodes{} and edges[]");
        std::map<int64_t, const JVal*> node_by_id;
        std::map<int64_t, int64_t> parent, canvas_parent;                              // EdgeKind::Input / EdgeKind::Canvas
        for (const auto& kv : nodes->o) node_by_id[std::atoll(kv.first.c_str())] = &kv.second;
        for (const JVal& e : edges->a) {
            const int64_t from = want_int(e, "from", "edge"), to = want_int(e, "to", "edge");
            const JVal* kind = e.get("kind");
            const bool is_canvas = kind && kind->t == JVal::Str && kind->s == "canvas";
            if (!kind || kind->t != JVal::Str || (kind->s != "input" && !is_canvas)) raise(kInvalidJson, "InvalidJson: edge kind is input or canvas");
            if (!node_by_id.count(from) || !node_by_id.count(to)) raise(kGraphInvalid, "GraphInvalid: edge names a missing node");
            auto& slot = is_canvas ? canvas_parent : parent;
            if (slot.count(to)) raise(kGraphInvalid, "GraphInvalid: node %lld has two %s edges", static_cast<long long>(to), is_canvas ? "canvas" : "input");
            slot[to] = from;
        }
        std::map<int64_t, FramePtr> done;
        std::map<int64_t, int> state;                                                  // 1 = on the stack (cycle check)
        // nodes that mutate a parent's frame in place (their input, or the canvas they draw on) must not see a frame
        // another consumer still needs: every such node with a shared parent works on its own copy
        std::map<int64_t, int> consumers;
        for (const auto& kv : parent) ++consumers[kv.second];
        for (const auto& kv : canvas_parent) ++consumers[kv.second];
        // "shared" is a property of the FRAME, not of the edge: a node that disappears (a resample_2d that has nothing to do, a
        // watermark on too small a canvas: delete_node_and_snap_together) hands its input on, and its consumer then reads a
        // frame the disappeared node's siblings still need
        std::map<const Frame*, bool> frame_shared;
        std::function<FramePtr(int64_t)> eval = [&](int64_t id) -> FramePtr {
            auto it = done.find(id);
            if (it != done.end()) return it->second;
            if (state[id] == 1) raise(kGraphInvalid, "GraphInvalid: cycle through node %lld", static_cast<long long>(id));
            state[id] = 1;
            std::string name;
            const JVal* params;
            node_of(*node_by_id[id], &name, &params);
            FramePtr in, canvas;
            bool in_shared = false;
            auto pit = parent.find(id);
            if (pit != parent.end()) {
                in = eval(pit->second);
                in_shared = in && frame_shared[in.get()];
                if (in && in_shared && mutates_input(name)) { in = clone(in, true); in_shared = false; }
            }
            auto cit = canvas_parent.find(id);
            if (cit != canvas_parent.end()) {
                canvas = eval(cit->second);
                if (canvas && frame_shared[canvas.get()]) canvas = clone(canvas);
            }
            const FramePtr given = in;
            lazy_decode = consumers[id] == 1;
            FramePtr out = run_node(name, *params, in, canvas, in_shared);
            if (out) frame_shared[out.get()] = consumers[id] > 1 || (out == given && in_shared);
            state[id] = 2;
            done[id] = out;
            return out;
        };
        for (const auto& kv : node_by_id) eval(kv.first);
    }
};

// Build001.io (imageflow_types/src/lib.rs:1433-1456, 1577-1581): placeholder / output_buffer need the buffers registered
// through the ABI; bytes_hex and base_64 carry the bytes inline.
void add_io_from_json(imageflow_context* c, const JVal& ios) {
    if (ios.t != JVal::Arr) raise(kInvalidJson, "InvalidJson: io must be an array");
    for (const JVal& o : ios.a) {
        const int32_t id = static_cast<int32_t>(want_int(o, "io_id", "io"));
        const JVal* dir = o.get("direction");
        const JVal* io = o.get("io");
        if (!dir || dir->t != JVal::Str || !io) raise(kInvalidJson, "InvalidJson: io entries need direction and io");
        const bool out = dir->s == "out";
        if (io->t == JVal::Str && io->s == "placeholder") {
            if (!c->io.count(id)) raise(kArgumentInvalid, "InvalidArgument: io_id %d is a placeholder but no buffer was added for it", id);
            continue;
        }
        if (c->io.count(id)) raise(kArgumentInvalid, "InvalidArgument: io_id %d is already in use", id);
        Io e;
        e.is_output = out;
        if (io->t == JVal::Str && (io->s == "output_buffer" || io->s == "output_base_64")) {
            if (!out) raise(kInvalidJson, "InvalidJson: %s on an input", io->s.c_str());
            e.out_base64 = io->s == "output_base_64";
        } else if (const JVal* arr = io->get("byte_array")) {                          // IoEnum::ByteArray(Vec<u8>)
            if (out || arr->t != JVal::Arr) raise(kInvalidJson, "InvalidJson: bad byte_array");
            e.owned.reserve(arr->a.size());
            for (const JVal& v : arr->a) {
                if (v.t != JVal::Num || v.n < 0 || v.n > 255 || v.n != std::floor(v.n)) raise(kInvalidJson, "InvalidJson: byte_array holds integers 0..255");
                e.owned.push_back(static_cast<uint8_t>(v.n));
            }
            e.in_len = e.owned.size();
        }
        else if (const JVal* hex = io->get("bytes_hex")) {
            if (out || hex->t != JVal::Str || (hex->s.size() & 1)) raise(kInvalidJson, "InvalidJson: bad bytes_hex");
            for (size_t i = 0; i < hex->s.size(); i += 2) e.owned.push_back(static_cast<uint8_t>(std::strtoul(hex->s.substr(i, 2).c_str(), nullptr, 16)));
            e.in_len = e.owned.size();
        } else if (const JVal* b64 = io->get("base_64")) {
            if (out || b64->t != JVal::Str) raise(kInvalidJson, "InvalidJson: bad base_64");
            uint32_t acc = 0;
            int bits = 0;
            for (unsigned char ch : b64->s) {
                int v = ch >= 'A' && ch <= 'Z' ? ch - 'A' : ch >= 'a' && ch <= 'z' ? ch - 'a' + 26 : ch >= '0' && ch <= '9' ? ch - '0' + 52 : ch == '+' ? 62 : ch == '/' ? 63 : -1;
                if (v < 0) continue;
                acc = (acc << 6) | static_cast<uint32_t>(v); bits += 6;
                if (bits >= 8) { bits -= 8; e.owned.push_back(static_cast<uint8_t>(acc >> bits)); }
            }
            e.in_len = e.owned.size();
        } else raise(kActionNotSupported, "ActionNotSupported: io kind (this shim: placeholder, output_buffer, output_base_64, bytes_hex, base_64, byte_array)");
        auto& slot = c->io[id] = std::move(e);
        if (!slot.is_output) slot.in = slot.owned.data();
    }
}

// s::ResultBytes (imageflow_types/src/lib.rs): "elsewhere" for output buffers, {"base_64": ...} for IoEnum::OutputBase64
std::string result_bytes(const Job& job, int32_t io_id) {
    auto it = job.c->io.find(io_id);
    if (it == job.c->io.end() || !it->second.out_base64) return "\"elsewhere\"";
    static const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const std::vector<uint8_t>& b = it->second.owned;
    std::string o = "{\"base_64\": \"";
    o.reserve(o.size() + (b.size() + 2) / 3 * 4 + 4);
    for (size_t i = 0; i < b.size(); i += 3) {
        const uint32_t n = (static_cast<uint32_t>(b[i]) << 16) | (i + 1 < b.size() ? static_cast<uint32_t>(b[i + 1]) << 8 : 0u) | (i + 2 < b.size() ? b[i + 2] : 0u);
        o.push_back(kB64[(n >> 18) & 63]); o.push_back(kB64[(n >> 12) & 63]);
        o.push_back(i + 1 < b.size() ? kB64[(n >> 6) & 63] : '='); o.push_back(i + 2 < b.size() ? kB64[n & 63] : '=');
    }
    return o + "\"}";
}

std::string job_result_json(const Job& job, const char* key) {
    std::string s = "{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"" + std::string(key) + "\": {\n      \"encodes\": [";
    for (size_t i = 0; i < job.encodes.size(); ++i) {
        const EncodeRecord& e = job.encodes[i];
        s += std::string(i ? "," : "") + "\n        {\"preferred_mime_type\": \"" + e.mime + "\", \"preferred_extension\": \"" + e.ext + "\", \"io_id\": " +
             std::to_string(e.io_id) + ", \"w\": " + std::to_string(e.w) + ", \"h\": " + std::to_string(e.h) + ", \"bytes\": " + result_bytes(job, e.io_id) + "}";
    }
    s += "\n      ],\n      \"decodes\": [";
    for (size_t i = 0; i < job.decodes.size(); ++i) {
        const DecodeRecord& d = job.decodes[i];
        s += std::string(i ? "," : "") + "\n        {\"preferred_mime_type\": \"" + d.mime + "\", \"preferred_extension\": \"" + d.ext + "\", \"io_id\": " +
             std::to_string(d.io_id) + ", \"w\": " + std::to_string(d.w) + ", \"h\": " + std::to_string(d.h) + "}";
    }
    // s::BuildPerformance {frames: [FramePerformance {nodes: [NodePerf {wall_microseconds, name}], wall_microseconds,
    // overhead_microseconds}]} (imageflow_types/src/lib.rs:1999-2016), filled as Engine::execute_many does
    // (flow/execution_engine.rs:179-199: nodes sorted by cost, largest first).  `gpu_microseconds` is an added field:
    // the time the device spent between the node's first and last launch (hipEvents).
    std::vector<NodePerf> nodes = job.perf;
    std::stable_sort(nodes.begin(), nodes.end(), [](const NodePerf& a, const NodePerf& b) { return a.wall_ns > b.wall_ns; });
    uint64_t total_node_ns = 0;
    for (const NodePerf& n : job.perf) total_node_ns += n.wall_ns;
    const int64_t total_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - job.t_start).count();
    s += "\n      ],\n      \"performance\": {\"frames\": [{\"nodes\": [";
    for (size_t i = 0; i < nodes.size(); ++i)
        s += std::string(i ? ", " : "") + "{\"wall_microseconds\": " + std::to_string(static_cast<uint64_t>(std::llround(static_cast<double>(nodes[i].wall_ns) / 1000.0))) +
             ", \"name\": \"" + nodes[i].name + "\", \"gpu_microseconds\": " + std::to_string(static_cast<uint64_t>(std::llround(static_cast<double>(nodes[i].gpu_ms) * 1000.0))) + "}";
    s += "], \"wall_microseconds\": " + std::to_string(static_cast<uint64_t>(std::llround(static_cast<double>(total_ns) / 1000.0))) +
         ", \"overhead_microseconds\": " + std::to_string(static_cast<int64_t>(std::llround(static_cast<double>(total_ns - static_cast<int64_t>(total_node_ns)) / 1000.0))) + "}]}\n    }\n  }\n}";
    return s;
}

// ExecutionSecurity overrides a job may carry (imageflow_core/src/context.rs:640-653)
void parse_security(const JVal* sec, Security* out) {
    if (!sec || sec->t != JVal::Obj) return;
    auto limit = [&](const char* key, SizeLimit* l) {
        const JVal* v = sec->get(key);
        if (!v || v->t != JVal::Obj) return;
        l->w = want_u32(*v, "w", key); l->h = want_u32(*v, "h", key);
        const JVal* mp = v->get("megapixels");
        if (!mp || mp->t != JVal::Num) raise(kInvalidJson, "InvalidJson: %s.megapixels must be a number", key);
        l->megapixels = static_cast<float>(mp->n);
    };
    limit("max_decode_size", &out->max_decode_size);
    limit("max_frame_size", &out->max_frame_size);
}

[[noreturn]] void abort_null_context() {                          // imageflow_abi/src/lib.rs:309-325
    fprintf(stderr, "Null context pointer provided. Terminating process.\n");
    std::abort();
}
#define CTX_OR_ABORT(c) do { if (!(c)) abort_null_context(); } while (0)

}  // namespace

// ==================================================================================================================
extern "C" {

bool imageflow_abi_compatible(uint32_t major, uint32_t minor) { return major == IMAGEFLOW_ABI_VER_MAJOR && minor <= IMAGEFLOW_ABI_VER_MINOR; }
uint32_t imageflow_abi_version_major(void) { return IMAGEFLOW_ABI_VER_MAJOR; }
uint32_t imageflow_abi_version_minor(void) { return IMAGEFLOW_ABI_VER_MINOR; }

struct imageflow_context* imageflow_context_create(uint32_t major, uint32_t minor) {       // lib.rs:430
    if (!imageflow_abi_compatible(major, minor)) return nullptr;
    try {
        imageflow_context* c = new imageflow_context;
        if (g_spread_contexts.load(std::memory_order_relaxed)) {
            const std::vector<int>& devs = usable_devices();
            if (!devs.empty()) c->device.store(devs[g_context_counter.fetch_add(1, std::memory_order_relaxed) % devs.size()], std::memory_order_relaxed);
        }
        return c;
    } catch (...) { return nullptr; }
}
bool imageflow_context_begin_terminate(struct imageflow_context* c) { CTX_OR_ABORT(c); return true; }
void imageflow_context_destroy(struct imageflow_context* c) { delete c; }

bool imageflow_context_has_error(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return c->err_cat != kOk; }
int32_t imageflow_context_error_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return c->err_cat; }
int32_t imageflow_context_error_as_exit_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return exit_code(c->err_cat); }
int32_t imageflow_context_error_as_http_code(struct imageflow_context* c) { CTX_OR_ABORT(c); std::lock_guard<std::mutex> lk(c->mu); return http_code(c->err_cat); }
// OutwardErrorBuffer::recoverable / try_clear (imageflow_core/src/errors.rs:953-969) over FlowError::recoverable, which
// is `false` for every error today (:656-658): "recoverable" is true only while there is no error, and an error, once
// set, is never cleared -- a caller that wants to go on creates a new context.
bool imageflow_context_error_recoverable(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    return c->err_cat == kOk;
}
bool imageflow_context_error_try_clear(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    return c->err_cat == kOk;
}
// lib.rs:744: print the error to stderr and exit with its exit code; returns false when there is none
bool imageflow_context_print_and_exit_if_error(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    int cat;
    std::string msg;
    { std::lock_guard<std::mutex> lk(c->mu); cat = c->err_cat; msg = c->err_msg; }
    if (cat == kOk) return false;
    fprintf(stderr, "%s\n", msg.c_str());
    std::exit(exit_code(cat));
}
// lib.rs:878: the one call meant for another thread while a job runs -- it takes no lock
void imageflow_context_request_cancellation(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    c->cancel.store(true, std::memory_order_relaxed);
}
// Context::request_cancellation_after_n_polls[_remaining] (context.rs:96-104,167-172; debug builds of the reference): the
// hook its own test_job_with_cancellation_at_every_point uses (imageflow_abi/src/lib.rs:1669-1743)
void ifhip_shim_request_cancellation_after_n_polls(struct imageflow_context* c, int64_t polls) {
    CTX_OR_ABORT(c);
    c->poll_countdown.store(polls, std::memory_order_seq_cst);
}
int ifhip_shim_process_constraint(const char* mode, int32_t source_w, int32_t source_h, int64_t w, int64_t h, int has_gravity, float gravity_x,
                                  float gravity_y, uint32_t* crop, int32_t* scale_to, uint32_t* pad, int32_t* canvas, int* flags) {
    if (!mode || !crop || !scale_to || !pad || !canvas || !flags) return 2;
    const int m = ifhip::constraint_mode_from_name(mode);
    if (m < 0) return 2;
    ifhip::ConstraintLayout l;
    std::string err;
    if (!ifhip::process_constraint(m, source_w, source_h, w, h, has_gravity != 0, gravity_x, gravity_y, &l, &err)) return 1;
    std::memcpy(crop, l.crop, sizeof l.crop);
    std::memcpy(pad, l.pad, sizeof l.pad);
    scale_to[0] = l.scale_w; scale_to[1] = l.scale_h;
    canvas[0] = l.canvas_w; canvas[1] = l.canvas_h;
    *flags = (l.has_crop ? 1 : 0) | (l.has_pad ? 2 : 0);
    return 0;
}
int64_t ifhip_shim_cancellation_polls_remaining(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    return c->poll_countdown.load(std::memory_order_seq_cst);
}
int64_t ifhip_shim_fused_decode_resamples(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    return c->fused_decode_resamples.load(std::memory_order_relaxed);
}
int64_t ifhip_shim_coalesced_decodes(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    return c->coalesced_decodes.load(std::memory_order_relaxed);
}
// Contexts over the node's GPUs (see DeviceScope): enable != 0 -> contexts created from now on take devices round-robin
void ifhip_shim_spread_contexts(int enable) { g_spread_contexts.store(enable ? 1 : 0, std::memory_order_relaxed); }
// bind one context to a device ordinal (-1: follow the calling thread again); false and an error on the context when the
// ordinal is not a usable device
bool ifhip_shim_context_set_device(struct imageflow_context* c, int ordinal) {
    CTX_OR_ABORT(c);
    const std::vector<int>& devs = usable_devices();
    if (ordinal != -1 && std::find(devs.begin(), devs.end(), ordinal) == devs.end()) {
        std::lock_guard<std::mutex> lk(c->mu);
        c->set_error(kArgumentInvalid, "InvalidArgument: device ordinal " + std::to_string(ordinal) + " is not a usable gfx950 device");
        return false;
    }
    c->device.store(ordinal, std::memory_order_relaxed);
    return true;
}
int ifhip_shim_context_device(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    return c->device.load(std::memory_order_relaxed);
}
int64_t ifhip_shim_device_coded_files(struct imageflow_context* c) {
    CTX_OR_ABORT(c);
    return c->device_coded_files.load(std::memory_order_relaxed);
}
bool imageflow_context_error_write_to_buffer(struct imageflow_context* c, char* buffer, size_t buffer_length, size_t* bytes_written) {   // lib.rs:684
    CTX_OR_ABORT(c);
    if (!buffer || buffer_length == 0 || (buffer_length >> (sizeof(size_t) * 8 - 1))) { if (bytes_written) *bytes_written = 0; return false; }
    std::lock_guard<std::mutex> lk(c->mu);
    const std::string& m = c->err_msg;
    static const char kTrunc[] = "\n[truncated]\n";
    bool whole = m.size() + 1 <= buffer_length;
    size_t n;
    if (whole) { n = m.size(); std::memcpy(buffer, m.data(), n); }
    else {
        const size_t t = sizeof kTrunc - 1;
        const size_t keep = buffer_length > t + 1 ? buffer_length - 1 - t : 0;
        std::memcpy(buffer, m.data(), keep);
        const size_t tn = std::min(t, buffer_length - 1 - keep);
        std::memcpy(buffer + keep, kTrunc, tn);
        n = keep + tn;
    }
    buffer[n] = 0;
    if (bytes_written) *bytes_written = n;
    return whole;
}

bool imageflow_context_add_input_buffer(struct imageflow_context* c, int32_t io_id, const uint8_t* buffer, size_t len, imageflow_lifetime lifetime) {   // lib.rs:1137
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!buffer) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'buffer' is null."); return false; }
    if (len >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: buffer_byte_count has its leading bit set"); return false; }
    if (c->io.count(io_id)) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is already in use"); return false; }
    Io e;
    if (lifetime == imageflow_lifetime_lifetime_outlives_context) { e.in = buffer; e.in_len = len; }
    else { e.owned.assign(buffer, buffer + len); e.in_len = len; }
    auto& slot = c->io[io_id] = std::move(e);
    if (lifetime != imageflow_lifetime_lifetime_outlives_context) slot.in = slot.owned.data();
    return true;
}
bool imageflow_context_add_output_buffer(struct imageflow_context* c, int32_t io_id) {       // lib.rs:1224
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->io.count(io_id)) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is already in use"); return false; }
    Io e;
    e.is_output = true;
    c->io[io_id] = std::move(e);
    return true;
}
bool imageflow_context_get_output_buffer_by_id(struct imageflow_context* c, int32_t io_id, const uint8_t** result_buffer, size_t* result_buffer_length) {   // lib.rs:1272
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!result_buffer || !result_buffer_length) { c->set_error(kArgumentInvalid, "NullArgument: result pointers are null"); return false; }
    auto it = c->io.find(io_id);
    if (it == c->io.end() || !it->second.is_output) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is not an output buffer"); return false; }
    if (it->second.out_state == OutState::Taken) { c->set_error(kArgumentInvalid, "InvalidArgument: Output buffer for io_id " + std::to_string(io_id) + " has already been taken"); return false; }
    it->second.out_state = OutState::Lent;                          // codecs/mod.rs:602-626: a lent pointer blocks take()
    *result_buffer = it->second.owned.data();
    *result_buffer_length = it->second.owned.size();
    return true;
}
// lib.rs:1335: the bytes move out of the context into a heap block the caller frees with imageflow_buffer_free
bool imageflow_context_take_output_buffer(struct imageflow_context* c, int32_t io_id, uint8_t** result_buffer, size_t* result_buffer_length) {
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!result_buffer) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'result_buffer' is null."); return false; }
    if (!result_buffer_length) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'result_buffer_length' is null."); return false; }
    auto it = c->io.find(io_id);
    if (it == c->io.end() || !it->second.is_output) { c->set_error(kArgumentInvalid, "InvalidArgument: io_id " + std::to_string(io_id) + " is not an output buffer"); return false; }
    Io& o = it->second;
    if (o.out_state == OutState::Lent) { c->set_error(kArgumentInvalid, "InvalidArgument: Cannot take output buffer for io_id " + std::to_string(io_id) + ": a raw pointer was already lent out"); return false; }   // codecs/mod.rs:565-571
    if (o.out_state == OutState::Taken) { c->set_error(kArgumentInvalid, "InvalidArgument: Output buffer for io_id " + std::to_string(io_id) + " has already been taken"); return false; }   // :572-578
    uint8_t* p = static_cast<uint8_t*>(std::malloc(o.owned.size() ? o.owned.size() : 1));
    if (!p) { c->set_error(kOutOfMemory, "AllocationFailed: output buffer"); return false; }
    std::memcpy(p, o.owned.data(), o.owned.size());
    *result_buffer = p;
    *result_buffer_length = o.owned.size();
    std::vector<uint8_t>().swap(o.owned);
    o.out_state = OutState::Taken;
    return true;
}
// lib.rs:1385: NULL is a no-op; always true.  (`length` is what Rust's Box<[u8]> needs to rebuild the slice; malloc'd here.)
bool imageflow_buffer_free(uint8_t* buffer, size_t /*length*/) {
    std::free(buffer);
    return true;
}

const struct imageflow_json_response* imageflow_context_send_json(struct imageflow_context* c, const char* method, const uint8_t* json_buffer, size_t json_buffer_size) {   // lib.rs:944
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);                         // operations serialise per context (lib.rs:13-33)
    if (!method) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'method' is null."); return nullptr; }
    if (!json_buffer) { c->set_error(kArgumentInvalid, "NullArgument: The argument 'json_buffer' is null."); return nullptr; }
    if (json_buffer_size >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: Argument `json_buffer_size` likely came from a negative integer."); return nullptr; }
    try {
        const std::string m = method;
        if (m == "v1/get_version_info")
            return respond(c, 200, std::string("{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"version_info\": {\"long_version_string\": \"") +
                                       ifhip_version() + " (libimageflow ABI subset " + std::to_string(IMAGEFLOW_ABI_VER_MAJOR) + "." + std::to_string(IMAGEFLOW_ABI_VER_MINOR) + ")\"}\n  }\n}");
        const bool build = m == "v1/build" || m == "v0.1/build", execute = m == "v1/execute" || m == "v0.1/execute";
        const bool tell = m == "v1/tell_decoder" || m == "v0.1/tell_decoder", scaled_info = m == "v1/get_scaled_image_info";
        if (!build && !execute && !tell && !scaled_info && m != "v1/get_image_info" && m != "v0.1/get_image_info") {
            c->set_error(kArgumentInvalid, "InvalidMessageEndpoint: " + m);
            return respond(c, 404, "{\n  \"success\": \"false\",\n  \"code\": 404,\n  \"message\": \"Endpoint name not understood\"}");   // json/mod.rs:158-168
        }
        if (json_buffer_size > 64u * 1024u * 1024u) raise(kArgumentInvalid, "SizeLimitExceeded: JSON payload exceeds max_json_bytes");   // ExecutionSecurity::max_json_bytes
        const JVal root = parse_json(json_buffer, json_buffer_size);
        if (root.t != JVal::Obj) raise(kInvalidJson, "InvalidJson: the message must be an object");
        // the context's device, then the job's stream and the promise its frees rest on (every release is preceded by quiesce())
        DeviceScope on_device(c->device.load(std::memory_order_relaxed));
        StreamLease lease;
        ifhip::QuiescedScope quiet;
        struct InFlight { InFlight() { g_jobs_in_flight.fetch_add(1, std::memory_order_relaxed); } ~InFlight() { g_jobs_in_flight.fetch_sub(1, std::memory_order_relaxed); } } in_flight;
        Job job{c};
        job.poll_cancel();
        if (tell) {                                                  // v1/tell_decoder {io_id, command} (json/endpoints/v1.rs:365-371)
            Io& in = job.input(static_cast<int32_t>(want_int(root, "io_id", "tell_decoder")));
            const JVal* cmd = root.get("command");
            if (!cmd) raise(kInvalidJson, "InvalidJson: tell_decoder needs a command");
            if (cmd->t == JVal::Obj && cmd->get("jpeg_downscale_hints")) {           // s::DecoderCommand::JpegDownscaleHints
                const JVal& j = *cmd->get("jpeg_downscale_hints");
                in.told = true;
                in.told_w = want_u32(j, "width", "jpeg_downscale_hints"); in.told_h = want_u32(j, "height", "jpeg_downscale_hints");
                const JVal* b = j.get("scale_luma_spatially");
                in.told_spatial = b && b->t == JVal::Bool && b->b;
                b = j.get("gamma_correct_for_srgb_during_spatial_luma_scaling");
                in.told_gamma = b && b->t == JVal::Bool && b->b;
            } else if (cmd->t == JVal::Str && cmd->s == "discard_color_profile") {
                in.told_discard_profile = true;                                      // the samples as they are: what this library does anyway
            } else if (!(cmd->t == JVal::Str && cmd->s == "ignore_color_profile_errors") &&
                       !(cmd->t == JVal::Obj && cmd->get("webp_decoder_hints"))) {   // profile errors (there is no CMS to fail) / WebP: nothing to act on
                raise(kInvalidJson, "InvalidJson: unknown decoder command");
            }
            return respond(c, 200, "{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {}\n}");                 // TellDecoderV1Response {} (v1.rs:177)
        }
        if (!build && !execute) {                                    // get_image_info / get_scaled_image_info {io_id}: header facts only
            Io& in = job.input(static_cast<int32_t>(want_int(root, "io_id", "get_image_info")));
            uint32_t w = 0, h = 0, bw[3], bh[3], ri = 0;
            int nc = 0;
            uint8_t hs[3], vs[3];
            uint16_t qt[192];
            int progressive = 0;
            (void)ri;
            check(ifhip_jpeg_frame_info(in.in, in.in_len, &w, &h, &nc, hs, vs, bw, bh, qt, &progressive));
            if (scaled_info && in.told && in.told_w > 0 && in.told_h > 0 && (w > in.told_w || h > in.told_h))   // MzDec::apply_downscaling on the told hints (:588-618)
                for (uint32_t i = 1; i < 8; ++i) {
                    if (i == 7) continue;
                    const uint32_t sw = static_cast<uint32_t>((static_cast<uint64_t>(w) * i + 7) / 8), sh = static_cast<uint32_t>((static_cast<uint64_t>(h) * i + 7) / 8);
                    if (sw >= in.told_w && sh >= in.told_h) { w = sw; h = sh; break; }
                }
            {   // get_unscaled_rotated_image_info / get_scaled_rotated_image_info (v1.rs:288,352 -> context.rs:486-501)
                const int flag = Job::exif_flag(in);
                if (flag >= 5 && flag <= 8) std::swap(w, h);
            }
            return respond(c, 200, "{\n  \"code\": 200,\n  \"success\": true,\n  \"message\": \"OK\",\n  \"data\": {\n    \"image_info\": {\"preferred_mime_type\": \"image/jpeg\", "
                                   "\"preferred_extension\": \"jpg\", \"image_width\": " + std::to_string(w) + ", \"image_height\": " + std::to_string(h) +
                                   ", \"frame_decodes_into\": \"bgr_32\"}\n  }\n}");
        }
        if (build) { if (const JVal* bc = root.get("builder_config")) parse_security(bc->get("security"), &job.sec); }
        else parse_security(root.get("security"), &job.sec);
        if (build) if (const JVal* ios = root.get("io")) add_io_from_json(c, *ios);
        const JVal* fw = root.get("framewise");
        if (!fw || fw->t != JVal::Obj) raise(kInvalidJson, "InvalidJson: missing framewise");
        job.run_framewise(*fw);
        job.settle_perf(true);
        return respond(c, 200, job_result_json(job, build ? "build_result" : "job_result"));
    } catch (const FlowErr& e) {
        return respond_error(c, e.cat, e.msg);
    } catch (const std::bad_alloc&) {
        return respond_error(c, kOutOfMemory, "AllocationFailed: host memory");
    } catch (const std::exception& e) {                              // the catch_unwind of lib.rs:973-1017
        c->set_error(kInternalError, std::string("InternalError: ") + e.what());
        return nullptr;
    }
}

bool imageflow_json_response_read(struct imageflow_context* c, const struct imageflow_json_response* r, int64_t* status, const uint8_t** buf, size_t* len) {   // lib.rs:783
    CTX_OR_ABORT(c);
    if (!r) { std::lock_guard<std::mutex> lk(c->mu); c->set_error(kArgumentInvalid, "NullArgument: The argument response_in is null."); return false; }
    if (status) *status = r->r.status;
    if (buf) *buf = reinterpret_cast<const uint8_t*>(r->r.json.data());
    if (len) *len = r->r.json.size();
    return true;
}
bool imageflow_json_response_destroy(struct imageflow_context* c, struct imageflow_json_response* r) {   // lib.rs:842
    CTX_OR_ABORT(c);
    if (!r) return true;
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->responses.begin(); it != c->responses.end(); ++it)
        if (it->get() == r) { c->responses.erase(it); return true; }
    return false;
}

void* imageflow_context_memory_allocate(struct imageflow_context* c, size_t bytes, const char* /*filename*/, int32_t /*line*/) {   // lib.rs:1424
    CTX_OR_ABORT(c);
    std::lock_guard<std::mutex> lk(c->mu);
    if (bytes >> (sizeof(size_t) * 8 - 1)) { c->set_error(kArgumentInvalid, "InvalidArgument: bytes has its leading bit set"); return nullptr; }
    try {
        c->allocations.emplace_back(new uint8_t[bytes ? bytes : 1]());
        return c->allocations.back().get();
    } catch (...) { c->set_error(kOutOfMemory, "AllocationFailed"); return nullptr; }
}
bool imageflow_context_memory_free(struct imageflow_context* c, void* p, const char* /*filename*/, int32_t /*line*/) {   // lib.rs:1484
    CTX_OR_ABORT(c);
    if (!p) return true;
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->allocations.begin(); it != c->allocations.end(); ++it)
        if (it->get() == p) { c->allocations.erase(it); return true; }
    return false;
}

}  // extern "C"
