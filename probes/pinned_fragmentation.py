#!/usr/bin/env python3
"""Round 6 (second half of the question): the copy engine's rate into a page-locked block that was pinned
AFTER a larger page-locked block had been freed.  bench.py's default run meets this between its legs (the 16-byte-record leg of
configs[3] at threshold 0.01 parks an 8 GB result block; the 8-byte-record leg behind it pins 5 GB anew): 134 instead of 78 ms per
step, "one run in three" ("The result pipeline").  Plain HIP through ctypes, no library of this repository:
  1. pin 5 GiB, time a 4 GiB device-to-host copy into it (fresh process: the baseline), free it;
  2. pin 8 GiB, touch it, free it; pin 5 GiB again, time the same copy;
  3. the same once more after pinning and freeing 2 x 8 GiB.
Run on the GPU box: `python3 probes/pinned_fragmentation.py`."""
import ctypes as C
import json
import time

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hip error %d" % (what, rc))


def pin(n):
    p = C.c_void_p()
    t0 = time.perf_counter()
    chk(hip.hipHostMalloc(C.byref(p), n, 0), "hipHostMalloc")
    return p, time.perf_counter() - t0


def d2h(dev, host, n, reps=4):
    out = []
    for _ in range(reps):
        chk(hip.hipDeviceSynchronize(), "sync")
        t0 = time.perf_counter()
        chk(hip.hipMemcpyAsync(host, dev, n, 2, None), "hipMemcpyAsync")
        chk(hip.hipDeviceSynchronize(), "sync")
        out.append(round(n / (time.perf_counter() - t0) / 1e9, 1))
    return out


def main():
    G = 1 << 30
    dev = C.c_void_p()
    chk(hip.hipMalloc(C.byref(dev), 4 * G), "hipMalloc")
    chk(hip.hipMemset(dev, 1, 4 * G), "hipMemset")
    rec = {}
    a, t_a = pin(5 * G)
    rec["fresh_5GiB"] = {"pin_s": round(t_a, 2), "d2h_GBps": d2h(dev, a, 4 * G)}
    chk(hip.hipHostFree(a), "hipHostFree")
    for tag, n_big in (("after_8GiB_pinned_and_freed", 1), ("after_two_more", 2)):
        for _ in range(n_big):
            b, t_b = pin(8 * G)
            C.memset(b, 2, 8 * G)   # the caller writes its results here
            chk(hip.hipHostFree(b), "hipHostFree")
        a, t_a = pin(5 * G)
        rec[tag] = {"pin_s": round(t_a, 2), "d2h_GBps": d2h(dev, a, 4 * G)}
        chk(hip.hipHostFree(a), "hipHostFree")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
