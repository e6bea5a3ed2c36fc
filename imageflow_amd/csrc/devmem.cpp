// devmem.cpp -- device / pinned memory for objects that live as long as ONE JOB (resample plans, JPEG stages, entropy
// handles, the frames of a v1/execute job), and the stream a thread's job runs on.
//
// imageflow runs one Context per thread (imageflow_abi/src/lib.rs:20-27) and a service runs thousands of jobs a second.
// hipMalloc / hipFree cost tens of microseconds each and hipFree waits for the whole device, so a job that allocates its
// coefficient planes, frames and stage scratch afresh serialises every other thread's job behind its frees (measured
// through the libimageflow ABI, round 4: 430 jobs/s with 1 thread and 380 with 64).  Here freed blocks go to size-class
// free lists per device and come back without touching the driver; the total kept is capped, beyond it blocks really go.
//
// What hipFree's device-wide wait protected -- a block handed out again while the previous owner's kernels still run --
// is kept: a plain cached_free waits for the device exactly like hipFree; a caller that HAS synchronised the stream its
// work ran on says so (QuiescedScope) and pays nothing.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace ifhip {
namespace {

constexpr size_t kMinClass = 256;
// What the free lists may hold (blocks handed out do not count).  The memory parked here is invisible to every other allocator
// on the device -- torch's caching allocator, another process -- so the default is what a few hundred concurrent thumbnail jobs
// recycle (a 4K decode job: 33 MB frame + 25 MB coefficient planes), not a share of the 288 GB; a service that runs nothing
// else on the device raises it (ifhip_cache_set_limits), anybody can give the lists back (ifhip_cache_trim).
std::atomic<size_t> g_keep_device_bytes{size_t(8) << 30};  // per device
std::atomic<size_t> g_keep_host_bytes{size_t(1) << 30};

// size classes: 2^k * {8..15} / 8 -- at most 12.5 % above the request
size_t size_class(size_t bytes) {
    if (bytes <= kMinClass) return kMinClass;
    size_t p = kMinClass;
    while (p * 2 < bytes) p *= 2;                                     // p < bytes <= 2p
    const size_t step = p / 8;
    return p + (bytes - p + step - 1) / step * step;
}

struct Cache {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_lists;                  // class -> blocks
    std::map<void*, size_t> live;                                     // block -> class (handed out)
    // blocks released "behind" a stream (cached_free_after): reusable at once by an allocation FOR THAT STREAM, by anybody
    // once the event recorded at the release has completed.  Counted in `kept`.
    struct Pending { void* p; size_t cls; hipEvent_t ev; hipStream_t st; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> spare_events;
    size_t kept = 0, live_bytes = 0;
    uint64_t hits = 0, driver_allocs = 0, driver_frees = 0, oom_flushes = 0, device_syncs = 0;   // (ifhip_cache_stats)
};
Cache& device_cache(int dev) {
    static std::mutex mu;
    static std::map<int, Cache*> caches;                              // (never destroyed: blocks outlive static destruction order)
    std::lock_guard<std::mutex> lk(mu);
    Cache*& c = caches[dev];
    if (!c) c = new Cache;
    return *c;
}
Cache& host_cache() { static Cache* c = new Cache; return *c; }

thread_local int t_quiesced = 0;
thread_local hipStream_t t_stream = nullptr;

}  // namespace

void quiesced_enter() { ++t_quiesced; }
void quiesced_leave() { --t_quiesced; }
void* thread_stream() { return t_stream; }

namespace {
// (c.mu held) pending releases whose stream has passed the release point move to the free lists; at most `budget` queries
void reap_pending(Cache& c, size_t budget) {
    for (size_t i = 0; i < c.pending.size() && budget > 0; --budget) {
        const hipError_t e = hipEventQuery(c.pending[i].ev);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); ++i; continue; }
        (void)hipGetLastError();                                      // (an error here: the stream is gone; its work is over either way)
        c.free_lists[c.pending[i].cls].push_back(c.pending[i].p);
        c.spare_events.push_back(c.pending[i].ev);
        c.pending[i] = c.pending.back();
        c.pending.pop_back();
    }
}
}  // namespace

int cached_malloc(void** out, size_t bytes) { return cached_malloc_for_stream(out, bytes, nullptr, false); }

// `for_stream`: the block will only be touched by work queued on `stream` from now on -- a block released behind that very
// stream (cached_free_after) can be handed out again without waiting for anything: stream order is the guarantee.
int cached_malloc_for_stream(void** out, size_t bytes, void* stream, bool for_stream) {
    *out = nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipErrorNoDevice);
    const size_t cls = size_class(bytes ? bytes : 1);
    Cache& c = device_cache(dev);
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (for_stream)
            for (size_t i = c.pending.size(); i-- > 0;)
                if (c.pending[i].cls == cls && c.pending[i].st == static_cast<hipStream_t>(stream)) {
                    *out = c.pending[i].p;
                    c.spare_events.push_back(c.pending[i].ev);
                    c.pending[i] = c.pending.back();
                    c.pending.pop_back();
                    break;
                }
        if (!*out) {
            auto it = c.free_lists.find(cls);
            if ((it == c.free_lists.end() || it->second.empty()) && !c.pending.empty()) { reap_pending(c, 16); it = c.free_lists.find(cls); }
            if (it != c.free_lists.end() && !it->second.empty()) { *out = it->second.back(); it->second.pop_back(); }
        }
        if (*out) {
            c.kept -= cls;
            c.live[*out] = cls;
            c.live_bytes += cls;
            ++c.hits;
            return 0;
        }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {                                            // out of memory with blocks parked in the cache: give them back, once
        (void)hipGetLastError();
        // ... the free lists and the blocks still behind a stream alike (up to the cache's cap of them): their release points
        // are waited for OUTSIDE the lock, a short wait against an allocation that would otherwise fail
        std::vector<void*> drop;
        std::vector<Cache::Pending> waiting;
        {
            std::lock_guard<std::mutex> lk(c.mu);
            for (auto& kv : c.free_lists) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            waiting.swap(c.pending);
            c.kept = 0;
            ++c.oom_flushes;
            c.driver_frees += drop.size() + waiting.size();
        }
        for (const Cache::Pending& q : waiting) { (void)hipEventSynchronize(q.ev); (void)hipGetLastError(); drop.push_back(q.p); }
        for (void* p : drop) (void)hipFree(p);
        if (!waiting.empty()) {
            std::lock_guard<std::mutex> lk(c.mu);
            for (const Cache::Pending& q : waiting) c.spare_events.push_back(q.ev);
        }
        e = hipMalloc(out, cls);
        if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return static_cast<int>(e); }
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[*out] = cls;
    c.live_bytes += cls;
    ++c.driver_allocs;
    return 0;
}

int cached_free(void* p) {
    if (!p) return 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipFree(p));
    // the block may be handed to another thread at once: nothing of the previous owner's work may still touch it
    const bool sync = t_quiesced == 0;
    if (sync) (void)hipDeviceSynchronize();
    Cache& c = device_cache(dev);
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (sync) ++c.device_syncs;
        auto it = c.live.find(p);
        if (it == c.live.end()) cls = 0;                              // not ours (another device's block, or a foreign pointer)
        else {
            cls = it->second;
            c.live.erase(it);
            c.live_bytes -= cls;
            if (c.kept + cls <= g_keep_device_bytes.load(std::memory_order_relaxed)) {
                c.free_lists[cls].push_back(p);
                c.kept += cls;
                return 0;
            }
        }
        ++c.driver_frees;
    }
    return static_cast<int>(hipFree(p));
}

// Release without a host wait: the block goes back once the work queued on `stream` so far has run (an event marks the
// point).  What hipFreeAsync does, with this cache's accounting -- and without the runtime pool's habit of giving memory back
// to the driver at every synchronisation (release threshold 0: the no-hint ABI job ran at 1 200 or 2 800 jobs/s depending on
// which way the pool fell in a given process, profiles/r5_abi_jobs_cfg4_bimodal_by_process_not_threads.txt).
int cached_free_after(void* p, void* stream) {
    if (!p) return 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipFree(p));
    Cache& c = device_cache(dev);
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (!c.spare_events.empty()) { ev = c.spare_events.back(); c.spare_events.pop_back(); }
    }
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return cached_free(p); }
    if (hipEventRecord(ev, static_cast<hipStream_t>(stream)) != hipSuccess) {
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(c.mu); c.spare_events.push_back(ev); }
        return cached_free(p);
    }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t cls = it->second;
            if (c.kept + cls <= g_keep_device_bytes.load(std::memory_order_relaxed)) {
                c.live.erase(it);
                c.live_bytes -= cls;
                c.pending.push_back({p, cls, ev, static_cast<hipStream_t>(stream)});
                c.kept += cls;
                if (c.pending.size() > 64) reap_pending(c, 8);
                return 0;
            }
        }
        c.spare_events.push_back(ev);
    }
    return cached_free(p);                                            // over the cap, or not ours: the waiting way
}

int cached_host_malloc(void** out, size_t bytes) {
    *out = nullptr;
    const size_t cls = size_class(bytes ? bytes : 1);
    Cache& c = host_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_lists.find(cls);
        if (it != c.free_lists.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            c.kept -= cls;
            c.live[*out] = cls;
            c.live_bytes += cls;
            ++c.hits;
            return 0;
        }
    }
    const hipError_t e = hipHostMalloc(out, cls, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return static_cast<int>(e); }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[*out] = cls;
    c.live_bytes += cls;
    ++c.driver_allocs;
    return 0;
}

int cached_host_free(void* p) {
    if (!p) return 0;
    Cache& c = host_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t cls = it->second;
            c.live.erase(it);
            c.live_bytes -= cls;
            if (c.kept + cls <= g_keep_host_bytes.load(std::memory_order_relaxed)) {
                c.free_lists[cls].push_back(p);
                c.kept += cls;
                return 0;
            }
        }
        ++c.driver_frees;
    }
    return static_cast<int>(hipHostFree(p));
}

namespace {
// give free-listed blocks back to the driver until at most `keep` bytes stay parked; largest classes first
template <typename FreeFn>
size_t trim_cache(Cache& c, size_t keep, FreeFn release) {
    std::vector<void*> drop;
    size_t dropped = 0;
    // releases still behind a stream are cache like the rest: taken out under the lock, waited for WITHOUT it (every
    // cached_malloc / cached_free of the device would stall behind the event waits otherwise), then free-listed
    std::vector<Cache::Pending> waiting;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        waiting.swap(c.pending);
    }
    for (const Cache::Pending& q : waiting) { (void)hipEventSynchronize(q.ev); (void)hipGetLastError(); }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (const Cache::Pending& q : waiting) {
            c.free_lists[q.cls].push_back(q.p);
            c.spare_events.push_back(q.ev);
        }
        for (auto it = c.free_lists.rbegin(); it != c.free_lists.rend() && c.kept > keep; ++it)
            while (!it->second.empty() && c.kept > keep) {
                drop.push_back(it->second.back());
                it->second.pop_back();
                c.kept -= it->first;
                dropped += it->first;
            }
        c.driver_frees += drop.size();
    }
    for (void* p : drop) release(p);
    return dropped;
}
}  // namespace

// "is this device a gfx950?" -- hipGetDeviceProperties fills a kilobyte struct through the driver; asked once per device
int require_gfx950(int* device_out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: no HIP device; this library has no CPU path");
    static std::mutex mu;
    static std::map<int, std::string> arch;
    std::string name;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = arch.find(dev);
        if (it == arch.end()) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(IFHIP_GPU_ERROR, "GpuError: hipGetDeviceProperties(%d) failed", dev);
            it = arch.emplace(dev, prop.gcnArchName).first;
        }
        name = it->second;
    }
    if (name.compare(0, 6, "gfx950") != 0)
        return fail(IFHIP_GPU_UNAVAILABLE, "GpuUnavailable: device %d is %s, this library is built for gfx950 only", dev, name.c_str());
    if (device_out) *device_out = dev;
    return IFHIP_OK;
}

// The library's one host-side wait.  hipStreamSynchronize SPINS in user space until the stream drains (the runtime's default,
// hipDeviceScheduleAuto; an event created with hipEventBlockingSync spins as well, ROCr polls ~200 us before it sleeps --
// both measured, profiles/r5_abi_jobs_host_cpu.txt): one busy host core per waiting thread.  That is the lowest latency while
// cores are idle, and a waste when they are not: with 40 jobs in flight in a container with a 16-CPU quota the runtime's wait
// for everybody keeps all 16 CPUs busy and the cgroup throttled, for 11 % fewer jobs per second than this policy makes with 6
// (profiles/r5_abi_jobs_wait_policy.txt).  So the wait is load-aware: the first few waiters (a quarter of the CPUs this process may use) spin in the runtime, the rest query the
// stream and sleep in between.
//   switch `wait`: "runtime" = always hipStreamSynchronize, "sleep" = always query + sleep; default = by load
//   switch `wait_spinners`: how many threads may spin at a time;  `wait_sleep_us`: the sleep between two queries (default 20)
namespace {
std::atomic<int> g_waiters{0};
int cpu_budget() {                                                    // CPUs this process may keep busy: affinity and cgroup quota
    static const int n = [] {
        int cpus = static_cast<int>(std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus > 0 ? cpus : CPU_COUNT(&set), CPU_COUNT(&set));
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {              // cgroup v2: "<quota|max> <period>"
            char q[32] = {0}; long long period = 0;
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && q[0] != 'm')
                cpus = std::min<long long>(cpus, std::max<long long>(1, (std::atoll(q) + period - 1) / period));
            std::fclose(f);
        } else {
            long long quota = -1, period = 0;                                    // cgroup v1
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
            if (quota > 0 && period > 0) cpus = std::min<long long>(cpus, std::max<long long>(1, (quota + period - 1) / period));
        }
        return std::max(1, cpus);
    }();
    return n;
}
}  // namespace

int wait_stream(void* hip_stream) {
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (!st) return static_cast<int>(hipStreamSynchronize(st));
    hipError_t e = hipStreamQuery(st);                                // often idle already
    if (e != hipErrorNotReady) return static_cast<int>(e);
    (void)hipGetLastError();                                          // (NotReady is recorded as the thread's last error)
    struct Waiting { int n; Waiting() : n(g_waiters.fetch_add(1, std::memory_order_relaxed) + 1) {} ~Waiting() { g_waiters.fetch_sub(1, std::memory_order_relaxed); } } me;
    const int spinners = std::max(1, cpu_budget() / 4);
    if (me.n <= spinners) return static_cast<int>(hipStreamSynchronize(st));
    constexpr long sleep_us = 20;
    for (;;) {
        std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
        e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return static_cast<int>(e);
        (void)hipGetLastError();
    }
}

// host <-> device copies of the create paths: on the calling thread's job stream, complete on return.  The host side is the
// caller's ordinary (pageable) memory, and for that hipMemcpyAsync is not asynchronous at all: the runtime stages the bytes
// itself and SPINS inside the call until they have moved -- 42 % of the host CPU of an ABI job was that spin, out of reach of
// wait_stream's load-aware policy (round 5, profiles/r5_abi_jobs_host_cpu_slots32.txt).  So the copies are staged here, through
// a pinned block of the library's cache, and the wait is ours.
namespace {
constexpr size_t kStageLimit = size_t(64) << 20;                       // larger copies: the runtime's own chunked path
}
int copy_to_device(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    void* pin = nullptr;
    if (bytes > kStageLimit || cached_host_malloc(&pin, bytes) != 0) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, t_stream);
        if (e == hipSuccess) e = static_cast<hipError_t>(wait_stream(t_stream));
        return static_cast<int>(e);
    }
    std::memcpy(pin, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, t_stream);
    const hipError_t w = static_cast<hipError_t>(wait_stream(t_stream));  // (also after a failed launch: nothing may still read `pin`)
    (void)cached_host_free(pin);
    return static_cast<int>(e != hipSuccess ? e : w);
}
// The same upload WITHOUT the wait: queued on the thread's stream from a pinned block that is the caller's to
// cached_host_free once that stream has been waited for (the caller has a wait coming anyway and folds this copy into it).
int stage_to_device(void* dst, const void* src, size_t bytes, void** pin_out) {
    *pin_out = nullptr;
    if (!bytes) return 0;
    void* pin = nullptr;
    if (bytes > kStageLimit || cached_host_malloc(&pin, bytes) != 0) return copy_to_device(dst, src, bytes);
    std::memcpy(pin, src, bytes);
    const hipError_t e = hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, t_stream);
    if (e != hipSuccess) { (void)static_cast<hipError_t>(wait_stream(t_stream)); (void)cached_host_free(pin); return static_cast<int>(e); }
    *pin_out = pin;
    return 0;
}
int copy_to_host(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    void* pin = nullptr;
    if (bytes > kStageLimit || cached_host_malloc(&pin, bytes) != 0) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, t_stream);
        if (e == hipSuccess) e = static_cast<hipError_t>(wait_stream(t_stream));
        return static_cast<int>(e);
    }
    hipError_t e = hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, t_stream);
    const hipError_t w = static_cast<hipError_t>(wait_stream(t_stream));
    if (e == hipSuccess && w == hipSuccess) std::memcpy(dst, pin, bytes);
    (void)cached_host_free(pin);
    return static_cast<int>(e != hipSuccess ? e : w);
}
int zero_device(void* dst, size_t bytes) {                            // ordered on the job stream (the stage's first launch follows on it)
    if (!bytes) return 0;
    hipError_t e = hipMemsetAsync(dst, 0, bytes, t_stream);
    if (e == hipSuccess) e = static_cast<hipError_t>(wait_stream(t_stream));
    return static_cast<int>(e);
}

}  // namespace ifhip

extern "C" void ifhip_set_thread_stream(void* hip_stream) { ifhip::t_stream = static_cast<hipStream_t>(hip_stream); }

extern "C" int ifhip_cache_set_limits(size_t device_bytes, size_t host_bytes) {
    ifhip::g_keep_device_bytes.store(device_bytes, std::memory_order_relaxed);
    ifhip::g_keep_host_bytes.store(host_bytes, std::memory_order_relaxed);
    return IFHIP_OK;
}

extern "C" int ifhip_cache_trim(size_t keep_device_bytes, size_t keep_host_bytes, size_t* released_device_bytes, size_t* released_host_bytes) {
    int dev = -1;
    size_t d = 0;
    // cached blocks are idle by construction (a block enters a list only behind a device-wide wait or a quiesced stream), so
    // releasing them needs no further wait here
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0) d = ifhip::trim_cache(ifhip::device_cache(dev), keep_device_bytes, [](void* p) { (void)hipFree(p); });
    const size_t h = ifhip::trim_cache(ifhip::host_cache(), keep_host_bytes, [](void* p) { (void)hipHostFree(p); });
    if (released_device_bytes) *released_device_bytes = d;
    if (released_host_bytes) *released_host_bytes = h;
    return IFHIP_OK;
}

extern "C" int ifhip_cache_stats(ifhip_cache_stats_t* out) {
    if (!out) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    std::memset(out, 0, sizeof *out);
    int dev = -1;
    auto fill = [](ifhip::Cache& c, uint64_t* v) {
        std::lock_guard<std::mutex> lk(c.mu);
        v[0] = c.hits; v[1] = c.driver_allocs; v[2] = c.driver_frees; v[3] = c.kept; v[4] = c.live_bytes; v[5] = c.live.size(); v[6] = c.oom_flushes; v[7] = c.device_syncs;
    };
    uint64_t v[8];
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0) {
        fill(ifhip::device_cache(dev), v);
        out->device_hits = v[0]; out->device_driver_allocs = v[1]; out->device_driver_frees = v[2]; out->device_bytes_cached = v[3];
        out->device_bytes_live = v[4]; out->device_blocks_live = v[5]; out->device_oom_flushes = v[6]; out->device_wide_syncs = v[7];
    }
    fill(ifhip::host_cache(), v);
    out->host_hits = v[0]; out->host_driver_allocs = v[1]; out->host_driver_frees = v[2]; out->host_bytes_cached = v[3];
    out->host_bytes_live = v[4]; out->host_blocks_live = v[5];
    out->device_limit_bytes = ifhip::g_keep_device_bytes.load(std::memory_order_relaxed);
    out->host_limit_bytes = ifhip::g_keep_host_bytes.load(std::memory_order_relaxed);
    return IFHIP_OK;
}
