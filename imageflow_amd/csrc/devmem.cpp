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

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace ifhip {
namespace {

constexpr size_t kMinClass = 256;
constexpr size_t kKeepDeviceBytes = size_t(24) << 30;      // per device (of 288 GB)
constexpr size_t kKeepHostBytes = size_t(2) << 30;

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
    size_t kept = 0;
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

int cached_malloc(void** out, size_t bytes) {
    *out = nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipErrorNoDevice);
    const size_t cls = size_class(bytes ? bytes : 1);
    Cache& c = device_cache(dev);
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_lists.find(cls);
        if (it != c.free_lists.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            c.kept -= cls;
            c.live[*out] = cls;
            return 0;
        }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {                                            // out of memory with blocks parked in the cache: give them back, once
        (void)hipGetLastError();
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(c.mu);
            for (auto& kv : c.free_lists) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            c.kept = 0;
        }
        for (void* p : drop) (void)hipFree(p);
        e = hipMalloc(out, cls);
        if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return static_cast<int>(e); }
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[*out] = cls;
    return 0;
}

int cached_free(void* p) {
    if (!p) return 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipFree(p));
    // the block may be handed to another thread at once: nothing of the previous owner's work may still touch it
    if (t_quiesced == 0) (void)hipDeviceSynchronize();
    Cache& c = device_cache(dev);
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) cls = 0;                              // not ours (another device's block, or a foreign pointer)
        else {
            cls = it->second;
            c.live.erase(it);
            if (c.kept + cls <= kKeepDeviceBytes) {
                c.free_lists[cls].push_back(p);
                c.kept += cls;
                return 0;
            }
        }
    }
    return static_cast<int>(hipFree(p));
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
            return 0;
        }
    }
    const hipError_t e = hipHostMalloc(out, cls, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return static_cast<int>(e); }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[*out] = cls;
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
            if (c.kept + cls <= kKeepHostBytes) {
                c.free_lists[cls].push_back(p);
                c.kept += cls;
                return 0;
            }
        }
    }
    return static_cast<int>(hipHostFree(p));
}

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

// host <-> device copies of the create paths: on the calling thread's job stream, complete on return
int copy_to_device(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, t_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t_stream);
    return static_cast<int>(e);
}
int copy_to_host(void* dst, const void* src, size_t bytes) {
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, t_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t_stream);
    return static_cast<int>(e);
}
int zero_device(void* dst, size_t bytes) {                            // ordered on the job stream (the stage's first launch follows on it)
    if (!bytes) return 0;
    hipError_t e = hipMemsetAsync(dst, 0, bytes, t_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t_stream);
    return static_cast<int>(e);
}

}  // namespace ifhip

extern "C" void ifhip_set_thread_stream(void* hip_stream) { ifhip::t_stream = static_cast<hipStream_t>(hip_stream); }
