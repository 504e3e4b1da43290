// Grow-only device and pinned-host scratch kept between calls of a one-shot entry point (sa_event_align_batch,
// sa_mea_batch), so that a caller feeding batches in a loop pays allocation once.  One instance per entry point and
// process; calls serialise on `mu`; the matching sa_*_release() returns the memory.
#ifndef SA_SCRATCH_H
#define SA_SCRATCH_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
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
// sa_hip.hip: device-side view of a finished batch for a downstream device step (per job: first pair in *pairs, number
// of pairs, number of events)
int sa_batch_device_view(sa_batch_t *b, const sa_pair_t **pairs, std::vector<long long> *first, std::vector<long long> *count,
                         std::vector<long long> *n_events, int *device);

// Host-side fan-out for per-read output building (fresh malloc'ed buffers are first-touch page faults: 60-100 MB of
// them per call are 5-10 ms on one thread).  fn(j) for j in [0, n), work handed out in blocks; SA_HOST_THREADS overrides
// the thread count (default: hardware threads, at most 16).
template <class F>
static inline void sa_parallel_for(size_t n, F fn) {
    unsigned want = std::thread::hardware_concurrency();
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
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < want; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
}

static inline size_t sa_up256(size_t x) { return (x + 255) & ~(size_t) 255; }

#endif
