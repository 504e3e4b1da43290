// Grow-only device and pinned-host scratch kept between calls of a one-shot entry point (sa_event_align_batch,
// sa_mea_batch), so that a caller feeding batches in a loop pays allocation once.  One instance per entry point and
// process; calls serialise on `mu`; the matching sa_*_release() returns the memory.
#ifndef SA_SCRATCH_H
#define SA_SCRATCH_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "sa_internal.h"

struct SaScratch {
    std::mutex mu;
    int device = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Slot {
        void **p;
        size_t *cap;
        bool pinned;
    };
    std::vector<Slot> slots;

    void release() {
        if (device >= 0) (void) hipSetDevice(device);
        for (Slot &s : slots) {
            if (*s.p) (void) (s.pinned ? hipHostFree(*s.p) : hipFree(*s.p));
            *s.p = nullptr;
            *s.cap = 0;
        }
        if (e0) (void) hipEventDestroy(e0);
        if (e1) (void) hipEventDestroy(e1);
        e0 = e1 = nullptr;
        device = -1;
    }
    int rebind(int dev) {
        if (device != dev) { release(); device = dev; }
        return hipSetDevice(dev) == hipSuccess ? SA_OK : SA_ENODEVICE;
    }
    int grow(void **p, size_t *cap, size_t bytes, int devno, bool pinned) {
        int rc = rebind(devno);
        if (rc) return rc;
        bool known = false;
        for (Slot &s : slots) known = known || s.p == p;
        if (!known) slots.push_back({p, cap, pinned});
        if (bytes <= *cap) return SA_OK;
        if (*p) (void) (pinned ? hipHostFree(*p) : hipFree(*p));
        *p = nullptr;
        *cap = 0;
        bytes += bytes / 8;
        hipError_t e = pinned ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
        if (e != hipSuccess) { *p = nullptr; (void) hipGetLastError(); return SA_ENOMEM; }
        *cap = bytes;
        return SA_OK;
    }
    int dev(void **p, size_t *cap, size_t bytes, int devno) { return grow(p, cap, bytes, devno, false); }
    int pin(void **p, size_t *cap, size_t bytes, int devno) { return grow(p, cap, bytes, devno, true); }
    int events() {
        if (e0) return SA_OK;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SA_ENODEVICE;
        return SA_OK;
    }
};
// Caching allocators for a batch's working storage.  A pipeline that sees every read once creates and destroys a batch
// per 2000 reads: hipMalloc / hipFree of its 25 GB (0.1 - 1 s per batch, measured), hipHostMalloc of the 200 MB pinned
// result buffer (40 ms) and their release (130 ms) cost ten times the kernels.  Blocks handed back are kept (per device
// and kind) and reused for requests they fit without wasting more than half of the block; sa_pool_release() returns
// everything, SA_POOL=0 disables the cache, SA_POOL_LIMIT_GB bounds what is held (default: 90 % of the device's memory, 32 GB pinned).
struct SaPool {
    enum Kind { DEVICE = 0, PINNED = 1 };
    struct Blk {
        void *p;
        size_t bytes;
        int dev;
    };
    std::mutex mu;
    std::vector<Blk> idle[2];
    std::unordered_map<void *, Blk> live[2];
    size_t held[2] = {0, 0};
    static bool enabled() {
        static const bool on = !(getenv("SA_POOL") && atoi(getenv("SA_POOL")) == 0);
        return on;
    }
    // sa_pool_configure(): limits set by the embedding caller (-1: not set); they win over the environment and the defaults
    static std::atomic<long long> &configured(int kind) {
        static std::atomic<long long> lim[2] = {{-1}, {-1}};
        return lim[kind];
    }
    static size_t limit(int kind) {
        const long long c = configured(kind).load();
        if (c >= 0) return (size_t) c;
        const char *e = getenv("SA_POOL_LIMIT_GB");
        if (e) return (size_t) (atof(e) * 1073741824.0);
        if (kind != DEVICE) return (size_t) 32 << 30;   // (pinning costs 0.25 ms per MB: three HDP batches of 4.3 GB of pairs stay cached)
        // device: whatever a destroyed batch held may stay parked (a 10k-event slice holds 170 GB of forward storage, and
        // hipFree + hipMalloc of it cost seconds); parked blocks are handed back when an allocation fails (get())
        static size_t dev_limit = 0;
        if (dev_limit == 0) {
            size_t free_b = 0, total_b = 0;
            dev_limit = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) ? (size_t) ((double) total_b * 0.9)
                                                                                        : (size_t) 96 << 30;
        }
        return dev_limit;
    }
    // Requests are rounded up to one of eight sizes per power of two: consecutive batches differ by a few per cent in every
    // array, and an exact-size cache would miss each time (a miss is a hipMalloc / hipHostMalloc, which also waits for the
    // batch that is running)
    static size_t round_up(size_t bytes) {
        if (bytes <= ((size_t) 1 << 16)) return ((bytes + 4095) / 4096) * 4096;
        size_t p2 = (size_t) 1 << 16;
        while (p2 * 2 <= bytes) p2 *= 2;
        const size_t step = p2 / 8;
        return ((bytes + step - 1) / step) * step;
    }
    static hipError_t raw_alloc(int kind, void **p, size_t bytes) {
        return kind == DEVICE ? hipMalloc(p, bytes) : hipHostMalloc(p, bytes, hipHostMallocDefault);
    }
    static void raw_free(int kind, void *p) { (void) (kind == DEVICE ? hipFree(p) : hipHostFree(p)); }
    // the current device must be `dev`
    hipError_t get(int kind, void **out, size_t bytes, int dev) {
        if (bytes == 0) bytes = 8;
        if (enabled()) bytes = round_up(bytes);
        if (enabled()) {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (size_t i = 0; i < idle[kind].size(); i++) {
                const Blk &b = idle[kind][i];
                if (b.dev != dev || b.bytes < bytes || b.bytes / 2 > bytes + (1 << 20)) continue;
                if (best < 0 || b.bytes < idle[kind][(size_t) best].bytes) best = (int) i;
            }
            if (best >= 0) {
                Blk b = idle[kind][(size_t) best];
                idle[kind].erase(idle[kind].begin() + best);
                held[kind] -= b.bytes;
                live[kind][b.p] = b;
                *out = b.p;
                return hipSuccess;
            }
        }
        // A pinned miss is expensive (hipHostMalloc: 0.25 ms per MB -- 100 ms for a batch's pair buffer) and stalls every thread of
        // the process that enters the HIP runtime meanwhile; a device miss is cheap by comparison (hipMalloc of 1.4 GB: < 0.1 ms
        // measured).  Consecutive batches of a stream ask for sizes within a few per cent of each other, sometimes on either side
        // of a size class: a new pinned block gets a sixteenth of headroom so that it also serves the neighbours' requests.
        const size_t want = bytes;
        if (enabled() && kind == PINNED && bytes >= ((size_t) 1 << 20)) bytes = round_up(bytes + bytes / 16);
        static const bool trace_pool = getenv("SA_TRACE") != nullptr;
        timespec ts0, ts1;
        if (trace_pool) clock_gettime(CLOCK_MONOTONIC, &ts0);
        hipError_t e = raw_alloc(kind, out, bytes);
        if (trace_pool) {
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            fprintf(stderr, "[trace] pool: new %s block of %.1f MB (asked %.1f MB): %.1f ms\n", kind == DEVICE ? "device" : "pinned",
                    bytes / 1048576.0, want / 1048576.0, (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6);
            if (kind == PINNED && bytes > ((size_t) 64 << 20)) {
                std::lock_guard<std::mutex> g(mu);
                fprintf(stderr, "[trace] pool: pinned idle (MB):");
                for (const Blk &q : idle[kind]) if (q.bytes > ((size_t) 16 << 20)) fprintf(stderr, " %.0f", q.bytes / 1048576.0);
                fprintf(stderr, " | live (MB):");
                for (const auto &kv : live[kind]) if (kv.second.bytes > ((size_t) 16 << 20)) fprintf(stderr, " %.0f", kv.second.bytes / 1048576.0);
                fprintf(stderr, "\n");
            }
        }
        if (e != hipSuccess && enabled()) {   // out of memory with blocks parked in the cache: give them back and retry
            (void) hipGetLastError();
            release(kind);
            bytes = want;
            e = raw_alloc(kind, out, bytes);
        }
        if (e == hipSuccess && enabled()) {
            std::lock_guard<std::mutex> g(mu);
            live[kind][*out] = Blk{*out, bytes, dev};
        }
        return e;
    }
    void put(int kind, void *p) {
        if (!p) return;
        Blk b{nullptr, 0, 0};
        bool known = false;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live[kind].find(p);
            if (it != live[kind].end()) {
                b = it->second;
                live[kind].erase(it);
                known = true;
                if (held[kind] + b.bytes <= limit(kind)) {
                    idle[kind].push_back(b);
                    held[kind] += b.bytes;
                    return;
                }
            }
        }
        (void) known;
        raw_free(kind, p);
    }
    // frees parked blocks (largest first) until what is held fits the current limit
    void trim(int kind) {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            const size_t lim = limit(kind);
            while (held[kind] > lim && !idle[kind].empty()) {
                size_t big = 0;
                for (size_t i = 1; i < idle[kind].size(); i++)
                    if (idle[kind][i].bytes > idle[kind][big].bytes) big = i;
                held[kind] -= idle[kind][big].bytes;
                drop.push_back(idle[kind][big].p);
                idle[kind].erase(idle[kind].begin() + (long) big);
            }
        }
        for (void *q : drop) raw_free(kind, q);
    }
    size_t idle_bytes(int kind, int dev) {
        std::lock_guard<std::mutex> g(mu);
        size_t n = 0;
        for (const Blk &b : idle[kind]) n += b.dev == dev ? b.bytes : 0;
        return n;
    }
    size_t live_bytes(int kind, int dev) {   // handed out and not yet returned
        std::lock_guard<std::mutex> g(mu);
        size_t n = 0;
        for (const auto &kv : live[kind]) n += kv.second.dev == dev ? kv.second.bytes : 0;
        return n;
    }
    void release(int kind) {
        std::vector<Blk> v;
        {
            std::lock_guard<std::mutex> g(mu);
            v.swap(idle[kind]);
            held[kind] = 0;
        }
        int cur = 0;
        (void) hipGetDevice(&cur);
        for (Blk &b : v) {
            (void) hipSetDevice(b.dev);
            raw_free(kind, b.p);
        }
        (void) hipSetDevice(cur);
    }
};
extern SaPool g_sa_pool;

// sa_hip.hip: device-side view of a finished batch for a downstream device step (per job: first pair in *pairs, number
// of pairs, number of events)
int sa_batch_device_view(sa_batch_t *b, const sa_pair16_t **pairs, std::vector<long long> *first, std::vector<long long> *count,
                         std::vector<long long> *n_events, int *device);

// Worker threads of the host fan-out below, started on first use and parked between calls: a fresh std::thread per worker and
// call cost 0.5-0.7 ms per fan-out at 13 workers, and sa_batch_create fans out five times (a third of its host time).  One
// fan-out at a time; callers from other threads queue up behind it.
struct SaWorkers {
    std::mutex call_mu;   // one fan-out at a time
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::vector<std::thread> th;
    const std::function<void()> *job = nullptr;
    unsigned long long gen = 0;
    unsigned take = 0, active = 0;   // workers that should join the current job / that have not finished it yet
    bool stop = false;
    void loop(unsigned idx) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void()> *f = nullptr;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (idx < take) f = job;
            }
            if (!f) continue;
            (*f)();
            std::lock_guard<std::mutex> g(mu);
            if (--active == 0) done_cv.notify_all();
        }
    }
    // runs `work` on `helpers` pool threads and on the caller; returns when all of them have returned
    void run(unsigned helpers, const std::function<void()> &work) {
        std::lock_guard<std::mutex> c(call_mu);
        {
            std::lock_guard<std::mutex> g(mu);
            while (th.size() < helpers) {
                const unsigned idx = (unsigned) th.size();
                th.emplace_back([this, idx] { loop(idx); });
            }
            job = &work; take = helpers; active = helpers; gen++;
        }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu);
        done_cv.wait(g, [&] { return active == 0; });
        job = nullptr;
    }
    ~SaWorkers() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        for (std::thread &t : th) t.join();
    }
};
extern SaWorkers g_sa_workers;

// Host-side fan-out for per-read output building (fresh malloc'ed buffers are first-touch page faults: 60-100 MB of
// them per call are 5-10 ms on one thread).  fn(j) for j in [0, n), work handed out in blocks; SA_HOST_THREADS overrides
// the thread count (default: hardware threads, at most 16).
template <class F>
static inline void sa_parallel_for(size_t n, F fn) {
    // default: hardware threads, at most 16 -- and at most the container's CPU quota minus three (the caller, the batch
    // runner thread and the HIP runtime's own threads need CPUs too; a fan-out that exceeds a cgroup quota gets the whole
    // process throttled until the end of the accounting period: measured 12.7 against 16-20 ms per pipelined batch)
    static const unsigned dflt = []() {
        unsigned w = std::thread::hardware_concurrency();
        w = w < 1 ? 1 : (w > 16 ? 16 : w);
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0}, per[64] = {0};
            if (fscanf(f, "%63s %63s", q, per) == 2 && strcmp(q, "max") != 0 && atof(per) > 0) {
                const double cpus = atof(q) / atof(per);
                const unsigned cap = cpus > 4.0 ? (unsigned) cpus - 3u : 1u;
                w = w > cap ? cap : w;
            }
            fclose(f);
        }
        return w;
    }();
    unsigned want = dflt;
    if (const char *e = getenv("SA_HOST_THREADS")) want = (unsigned) atoi(e);
    want = want < 1 ? 1 : (want > 16 ? 16 : want);
    const size_t block = 16;
    if (want == 1 || n <= block) {
        for (size_t j = 0; j < n; j++) fn(j);
        return;
    }
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t a = next.fetch_add(block);
            if (a >= n) return;
            const size_t b = a + block < n ? a + block : n;
            for (size_t j = a; j < b; j++) fn(j);
        }
    };
    static const bool parked = !(getenv("SA_WORKER_POOL") && atoi(getenv("SA_WORKER_POOL")) == 0);   // measurement hook
    if (parked) {
        g_sa_workers.run(want - 1, std::function<void()>(work));
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < want; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
}

static inline size_t sa_up256(size_t x) { return (x + 255) & ~(size_t) 255; }

#endif
