// bench_abi_jobs.cpp -- jobs per second through the OUTER boundary (libimageflow C ABI, include/imageflow_abi_subset.h):
// T host threads, one imageflow_context per job as the reference's guidance has it ("separate contexts per thread",
// imageflow_abi/src/lib.rs:20-27), every job = add_input_buffer + add_output_buffer + send_json("v1/execute") +
// take_output_buffer + buffer_free + context_destroy on a file already in host memory.  Reference bench shape:
// imageflow_core/benches/bench_graphics.rs:382-456 (one pipeline per iteration, wall clock over many).
//
//   bench_abi_jobs <libimageflow_hip.so> <file.jpg> <job.json> <threads> <seconds> [warmup_jobs_per_thread] [--spread]
//
// --spread: ifhip_shim_spread_contexts(1) first -- new contexts take the usable devices round-robin; the
// line then carries the jobs each device ran, and the run FAILS when a context lands on an ordinal that does not exist or,
// with more than one device, some device never got a job.
// The line also carries what the library's block cache did during the timed region (ifhip_cache_stats before / after):
// driver calls and device-wide waits per job, bytes cached and handed out at the end.
//
// Prints one JSON line: jobs/s, per-node means from the job results' `performance` block (wall and gpu microseconds),
// output bytes per job.  Development tool (tools/), built by tools/bench_abi_jobs.py with g++.
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/time.h>
#include <unistd.h>
#include <sys/resource.h>

#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct Api {
    void* (*context_create)(uint32_t, uint32_t);
    void (*context_destroy)(void*);
    bool (*add_input_buffer)(void*, int32_t, const uint8_t*, size_t, int);
    bool (*add_output_buffer)(void*, int32_t);
    const void* (*send_json)(void*, const char*, const uint8_t*, size_t);
    bool (*response_read)(void*, const void*, int64_t*, const uint8_t**, size_t*);
    bool (*response_destroy)(void*, const void*);
    bool (*take_output_buffer)(void*, int32_t, const uint8_t**, size_t*);
    bool (*buffer_free)(const uint8_t*, size_t);
    bool (*error_write)(void*, char*, size_t, size_t*);
    void (*spread)(int) = nullptr;                       // ifhip_shim_* / ifhip_*: this library's extensions, optional
    int (*context_device)(void*) = nullptr;
    int (*device_count)() = nullptr;
    int (*cache_stats)(void*) = nullptr;
};
// What the host spent: this process's CPU time (getrusage) and the container's CPU quota and throttling (cgroup v2 cpu.max /
// cpu.stat).  A quota of 16 CPUs with more than 16 runnable threads stops the WHOLE cgroup for the rest of each 100 ms period:
// the job rate then says nothing about the library (profiles/r5_abi_jobs_cpu_quota.txt).
struct HostCpu { double user_s = 0, sys_s = 0, quota_cpus = 0; long long nr_throttled = 0, throttled_usec = 0; };
static HostCpu host_cpu() {
    HostCpu h;
    rusage ru{};
    getrusage(RUSAGE_SELF, &ru);
    h.user_s = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6;
    h.sys_s = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0}; long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && q[0] != 'm') h.quota_cpus = std::atof(q) / period;
        std::fclose(f);
    }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.stat", "r")) {
        char k[64]; long long v;
        while (std::fscanf(f, "%63s %lld", k, &v) == 2) {
            if (!std::strcmp(k, "nr_throttled")) h.nr_throttled = v;
            if (!std::strcmp(k, "throttled_usec")) h.throttled_usec = v;
        }
        std::fclose(f);
    }
    return h;
}
// --- where the host CPU time goes: IFHIP_BENCH_SAMPLE=<file> samples the process on CPU time (ITIMER_PROF, 1 kHz: the signal lands on
// a thread in proportion to the CPU it burns) and writes one call stack per line as module+offset frames, leaf first.
// tools/sample_stacks.py symbolises and ranks them.
static constexpr int kMaxSamples = 120000, kDepth = 24;
static void* g_samples[kMaxSamples][kDepth];
static std::atomic<int> g_n_samples{0};
static std::atomic<bool> g_sampling{false};
static void on_prof(int) {
    if (!g_sampling.load(std::memory_order_relaxed)) return;
    const int i = g_n_samples.fetch_add(1, std::memory_order_relaxed);
    if (i >= kMaxSamples) return;
    int n = backtrace(g_samples[i], kDepth);
    for (; n < kDepth; ++n) g_samples[i][n] = nullptr;
}
// a crash inside the library (or the runtime, or a profiler's interposer) says where: module+offset frames on stderr
static void on_segv(int sig) {
    void* fr[40];
    const int n = backtrace(fr, 40);
    char line[600];
    for (int d = 0; d < n; ++d) {
        Dl_info di{};
        int m;
        if (dladdr(fr[d], &di) && di.dli_fname)
            m = std::snprintf(line, sizeof line, "crash frame %d: %s+0x%zx %s\n", d, di.dli_fname, static_cast<size_t>(static_cast<char*>(fr[d]) - static_cast<char*>(di.dli_fbase)), di.dli_sname ? di.dli_sname : "");
        else m = std::snprintf(line, sizeof line, "crash frame %d: ?\n", d);
        if (m > 0) (void)!write(2, line, static_cast<size_t>(m));
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
static void start_sampling() {
    void* warm[4]; (void)backtrace(warm, 4);                // loads libgcc's unwinder outside the handler
    struct sigaction sa{}; sa.sa_handler = on_prof; sa.sa_flags = SA_RESTART; sigaction(SIGPROF, &sa, nullptr);
    itimerval tv{{0, 1000}, {0, 1000}}; setitimer(ITIMER_PROF, &tv, nullptr);
}
static void write_samples(const char* path) {
    itimerval off{}; setitimer(ITIMER_PROF, &off, nullptr);
    FILE* f = std::fopen(path, "w");
    if (!f) return;
    const int n = std::min(g_n_samples.load(), kMaxSamples);
    for (int i = 0; i < n; ++i) {
        for (int d = 2; d < kDepth && g_samples[i][d]; ++d) {  // 0, 1: the handler and the signal trampoline
            Dl_info di{};
            if (dladdr(g_samples[i][d], &di) && di.dli_fname)
                std::fprintf(f, "%s%s+0x%zx", d > 2 ? ";" : "", di.dli_fname, static_cast<size_t>(static_cast<char*>(g_samples[i][d]) - static_cast<char*>(di.dli_fbase)));
            else std::fprintf(f, "%s?+0x0", d > 2 ? ";" : "");
        }
        std::fputc('\n', f);
    }
    std::fclose(f);
}
struct CacheStats {                                      // include/imageflow_hip.h ifhip_cache_stats_t
    uint64_t device_hits, device_driver_allocs, device_driver_frees, device_oom_flushes, device_wide_syncs;
    uint64_t device_bytes_cached, device_bytes_live, device_blocks_live, device_limit_bytes;
    uint64_t host_hits, host_driver_allocs, host_driver_frees;
    uint64_t host_bytes_cached, host_bytes_live, host_blocks_live, host_limit_bytes;
};

template <typename F>
static void load(void* h, const char* name, F* f) {
    *f = reinterpret_cast<F>(dlsym(h, name));
    if (!*f) { std::fprintf(stderr, "missing symbol %s\n", name); std::exit(2); }
}

static std::vector<uint8_t> read_file(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    std::fclose(f);
    return d;
}

struct NodeSum { double wall_us = 0, gpu_us = 0; uint64_t n = 0; };

// {"wall_microseconds": W, "name": "N", "gpu_microseconds": G} entries of performance.frames[].nodes[]
static void scan_nodes(const char* s, size_t len, std::map<std::string, NodeSum>* sums) {
    const std::string t(s, len);
    size_t at = t.find("\"nodes\"");
    if (at == std::string::npos) return;
    const size_t end = t.find(']', at);
    while (true) {
        const size_t w = t.find("\"wall_microseconds\": ", at);
        if (w == std::string::npos || w > end) break;
        const size_t nm = t.find("\"name\": \"", w), g = t.find("\"gpu_microseconds\": ", w);
        if (nm == std::string::npos || g == std::string::npos) break;
        const size_t nq = t.find('"', nm + 9);
        NodeSum& ns = (*sums)[t.substr(nm + 9, nq - nm - 9)];
        ns.wall_us += std::atof(t.c_str() + w + 21);
        ns.gpu_us += std::atof(t.c_str() + g + 20);
        ns.n += 1;
        at = g + 20;
    }
}

int main(int argc, char** argv) {
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    signal(SIGABRT, on_segv);
    if (argc < 6) { std::fprintf(stderr, "usage: %s lib file job.json threads seconds [warmup]\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    Api a;
    load(h, "imageflow_context_create", &a.context_create);
    load(h, "imageflow_context_destroy", &a.context_destroy);
    load(h, "imageflow_context_add_input_buffer", &a.add_input_buffer);
    load(h, "imageflow_context_add_output_buffer", &a.add_output_buffer);
    load(h, "imageflow_context_send_json", &a.send_json);
    load(h, "imageflow_json_response_read", &a.response_read);
    load(h, "imageflow_json_response_destroy", &a.response_destroy);
    load(h, "imageflow_context_take_output_buffer", &a.take_output_buffer);
    load(h, "imageflow_buffer_free", &a.buffer_free);
    load(h, "imageflow_context_error_write_to_buffer", &a.error_write);
    a.spread = reinterpret_cast<void (*)(int)>(dlsym(h, "ifhip_shim_spread_contexts"));
    a.context_device = reinterpret_cast<int (*)(void*)>(dlsym(h, "ifhip_shim_context_device"));
    a.device_count = reinterpret_cast<int (*)()>(dlsym(h, "ifhip_device_count"));
    a.cache_stats = reinterpret_cast<int (*)(void*)>(dlsym(h, "ifhip_cache_stats"));
    bool spread = false;
    for (int i = 6; i < argc; ++i) if (std::strcmp(argv[i], "--spread") == 0) spread = true;
    if (spread) {
        if (!a.spread || !a.context_device || !a.device_count) { std::fprintf(stderr, "--spread: the library has no ifhip_shim_spread_contexts\n"); return 2; }
        a.spread(1);
    }
    const int n_devices = a.device_count ? a.device_count() : 1;
    std::vector<std::atomic<uint64_t>> per_device(static_cast<size_t>(std::max(1, n_devices)));
    std::atomic<uint64_t> bad_device{0};
    // development switches of the library, as tools/ pass them: IFHIP_<SWITCH>=value in the environment -> ifhip_debug_set
    if (auto set = reinterpret_cast<int (*)(const char*, const char*)>(dlsym(h, "ifhip_debug_set"))) {
        extern char** environ;
        for (char** e = environ; *e; ++e) {
            std::string kv(*e);
            const size_t eq = kv.find('=');
            if (kv.compare(0, 6, "IFHIP_") != 0 || eq == std::string::npos || kv.compare(0, 9, "IFHIP_LIB") == 0) continue;
            std::string key = kv.substr(6, eq - 6);
            for (char& ch : key) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
            set(key.c_str(), kv.c_str() + eq + 1);
        }
    }
    const std::vector<uint8_t> file = read_file(argv[2]), job = read_file(argv[3]);
    const int threads = std::atoi(argv[4]);
    const double seconds = std::atof(argv[5]);
    const int warmup = argc > 6 && argv[6][0] != '-' ? std::atoi(argv[6]) : 3;

    std::atomic<bool> go{false}, stop{false};
    std::atomic<uint64_t> jobs{0}, out_bytes{0}, failures{0};
    std::mutex mu;
    std::map<std::string, NodeSum> sums;
    std::string first_error;

    auto one_job = [&](std::map<std::string, NodeSum>* local, bool count) {
        void* c = a.context_create(3, 2);
        if (!c) { failures++; return; }
        if (spread && count) {
            const int d = a.context_device(c);
            if (d < 0 || d >= n_devices) bad_device++;
            else per_device[static_cast<size_t>(d)]++;
        }
        bool ok = a.add_input_buffer(c, 0, file.data(), file.size(), 1) && a.add_output_buffer(c, 1);
        const void* r = ok ? a.send_json(c, "v1/execute", job.data(), job.size()) : nullptr;
        int64_t status = 0;
        const uint8_t* body = nullptr;
        size_t blen = 0;
        if (r && a.response_read(c, r, &status, &body, &blen) && status == 200) {
            if (count) scan_nodes(reinterpret_cast<const char*>(body), blen, local);
            const uint8_t* ob = nullptr;
            size_t on = 0;
            if (a.take_output_buffer(c, 1, &ob, &on)) {
                if (count) { out_bytes += on; jobs++; }
                a.buffer_free(ob, on);
            } else failures++;
        } else {
            failures++;
            std::lock_guard<std::mutex> lk(mu);
            if (first_error.empty()) {
                char buf[600];
                size_t n = 0;
                if (a.error_write(c, buf, sizeof buf, &n)) first_error.assign(buf, std::min(n, sizeof buf - 1));
                else if (body) first_error.assign(reinterpret_cast<const char*>(body), std::min<size_t>(blen, 500));
            }
        }
        if (r) a.response_destroy(c, r);
        a.context_destroy(c);
    };

    std::vector<std::thread> pool;
    std::atomic<int> ready{0};
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            std::map<std::string, NodeSum> local;
            for (int i = 0; i < warmup; ++i) one_job(&local, false);
            ready++;
            while (!go.load()) std::this_thread::yield();
            while (!stop.load()) one_job(&local, true);
            std::lock_guard<std::mutex> lk(mu);
            for (auto& kv : local) { NodeSum& s = sums[kv.first]; s.wall_us += kv.second.wall_us; s.gpu_us += kv.second.gpu_us; s.n += kv.second.n; }
        });
    while (ready.load() < threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    CacheStats cs0{}, cs1{};
    if (a.cache_stats) a.cache_stats(&cs0);
    const char* sample_path = std::getenv("IFHIP_BENCH_SAMPLE");
    if (sample_path) { start_sampling(); g_sampling = true; }
    const HostCpu hc0 = host_cpu();
    const auto t0 = std::chrono::steady_clock::now();
    go = true;
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& th : pool) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (a.cache_stats) a.cache_stats(&cs1);
    const HostCpu hc1 = host_cpu();
    if (sample_path) { g_sampling = false; write_samples(sample_path); }
    const uint64_t n = jobs.load();
    const double nj = static_cast<double>(std::max<uint64_t>(n, 1));
    char cache[700];
    std::snprintf(cache, sizeof cache,
                  "{\"device_hits_per_job\": %.2f, \"device_driver_allocs_per_job\": %.3f, \"device_driver_frees_per_job\": %.3f, \"device_wide_syncs_per_job\": %.3f, "
                  "\"device_oom_flushes\": %llu, \"device_MB_cached\": %.1f, \"device_MB_live\": %.1f, \"device_limit_MB\": %.0f, "
                  "\"host_hits_per_job\": %.2f, \"host_driver_allocs_per_job\": %.3f, \"host_driver_frees_per_job\": %.3f, \"host_MB_cached\": %.1f, \"host_MB_live\": %.1f}",
                  (cs1.device_hits - cs0.device_hits) / nj, (cs1.device_driver_allocs - cs0.device_driver_allocs) / nj, (cs1.device_driver_frees - cs0.device_driver_frees) / nj,
                  (cs1.device_wide_syncs - cs0.device_wide_syncs) / nj, static_cast<unsigned long long>(cs1.device_oom_flushes - cs0.device_oom_flushes),
                  cs1.device_bytes_cached / 1e6, cs1.device_bytes_live / 1e6, cs1.device_limit_bytes / 1e6,
                  (cs1.host_hits - cs0.host_hits) / nj, (cs1.host_driver_allocs - cs0.host_driver_allocs) / nj, (cs1.host_driver_frees - cs0.host_driver_frees) / nj,
                  cs1.host_bytes_cached / 1e6, cs1.host_bytes_live / 1e6);
    char host[300];
    std::snprintf(host, sizeof host, "{\"cpus_busy\": %.2f, \"user_cpus\": %.2f, \"sys_cpus\": %.2f, \"cpu_ms_per_job\": %.3f, \"cgroup_cpu_quota\": %.2f, "
                  "\"cgroup_throttled_periods\": %lld, \"cgroup_throttled_s\": %.3f}",
                  (hc1.user_s + hc1.sys_s - hc0.user_s - hc0.sys_s) / dt, (hc1.user_s - hc0.user_s) / dt, (hc1.sys_s - hc0.sys_s) / dt,
                  (hc1.user_s + hc1.sys_s - hc0.user_s - hc0.sys_s) / nj * 1e3, hc1.quota_cpus,
                  static_cast<long long>(hc1.nr_throttled - hc0.nr_throttled), (hc1.throttled_usec - hc0.throttled_usec) / 1e6);
    std::string devs = "[";
    bool unused_device = false;
    for (size_t d = 0; d < per_device.size(); ++d) {
        devs += (d ? ", " : "") + std::to_string(per_device[d].load());
        if (spread && n_devices > 1 && per_device[d].load() == 0) unused_device = true;
    }
    devs += "]";
    std::string nodes = "{";
    for (auto& kv : sums)
        nodes += (nodes.size() > 1 ? ", \"" : "\"") + kv.first + "\": {\"per_job\": " + std::to_string(static_cast<double>(kv.second.n) / std::max<uint64_t>(n, 1)) +
                 ", \"wall_us\": " + std::to_string(kv.second.wall_us / std::max<uint64_t>(kv.second.n, 1)) +
                 ", \"gpu_us\": " + std::to_string(kv.second.gpu_us / std::max<uint64_t>(kv.second.n, 1)) + "}";
    nodes += "}";
    for (char& ch : first_error) if (ch == '"' || ch == '\n' || ch == '\\') ch = ' ';
    std::printf("{\"threads\": %d, \"seconds\": %.3f, \"jobs\": %llu, \"jobs_per_s\": %.1f, \"ms_per_job_per_thread\": %.3f, \"failures\": %llu, "
                "\"output_bytes_per_job\": %llu, \"input_bytes\": %zu, \"devices\": %d, \"spread\": %s, \"jobs_per_device\": %s, \"contexts_on_unknown_devices\": %llu, "
                "\"host\": %s, \"cache\": %s, \"nodes\": %s, \"first_error\": \"%s\"}\n",
                threads, dt, static_cast<unsigned long long>(n), n / dt, n ? dt * threads / n * 1e3 : 0.0,
                static_cast<unsigned long long>(failures.load()), static_cast<unsigned long long>(n ? out_bytes.load() / n : 0), file.size(),
                n_devices, spread ? "true" : "false", spread ? devs.c_str() : "null", static_cast<unsigned long long>(bad_device.load()),
                host, a.cache_stats ? cache : "null", nodes.c_str(), first_error.c_str());
    return (failures.load() || bad_device.load() || unused_device) ? 1 : 0;
}
