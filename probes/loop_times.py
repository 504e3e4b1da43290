"""Where the host thread of the pipelined bench loop spends a step: create / start / wait / collect+close (ms, means over the
steady part), beside the step time.  usage: python probes/loop_times.py [depth] [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
def _cpulist(path):
    out = []
    for part in open(path).read().strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out
if os.environ.get("PIN_NODE"):   # PIN_NODE=0|1: run on the CPUs of one NUMA node (set before any thread starts)
    cpus = set(_cpulist("/sys/devices/system/node/node%s/cpulist" % os.environ["PIN_NODE"])) & os.sched_getaffinity(0)
    os.sched_setaffinity(0, cpus)
    print("pinned to node", os.environ["PIN_NODE"], len(cpus), "cpus")
import numpy as np
import signalalign_amd as sa
from signalalign_amd import synth
mp = os.path.join("tests", "golden", "models", "testModelR9.4_450bps.nucleotide.6mer.template.model")
alpha, k, t10, tab = synth.parse_model_table(mp)
pm = sa.Model.load(mp)
params = sa.default_params(threshold=0.01, expansion=50, trace_back=100)
sets = [[synth.make_read(i + q * 2000, 5000, alpha, k, tab) for i in range(2000)] for q in range(3)]
arrays = [sa.JobArray(js) for js in sets]
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except OSError:
        return 0, 0, 0
flying = []
thr_prev = throttled()
T = {"create": [], "start": [], "wait": [], "collect": []}
t_begin = None
for s in range(steps):
    if s == 8:
        t_begin = time.perf_counter()
    a = time.perf_counter()
    cur = sa.Batch(pm, params, arrays[s % 3], device=0)
    b = time.perf_counter()
    cur.start(); flying.append(cur)
    c = time.perf_counter()
    d = e = c
    if len(flying) >= depth:
        old = flying.pop(0); old.wait()
        d = time.perf_counter()
        old.n_pairs(0)
        d2 = time.perf_counter()
        old.close()
        e = time.perf_counter()
        if s >= 8: T.setdefault("n_pairs", []).append(d2 - d)
    thr = throttled()
    if thr[0] != thr_prev[0]:
        print("step %d: cgroup throttled %d time(s), %.1f ms; the step took %.1f ms (create %.1f, wait %.1f)"
              % (s, thr[0] - thr_prev[0], (thr[1] - thr_prev[1]) / 1e3, (e - a) * 1e3, (b - a) * 1e3, (d - c) * 1e3))
    thr_prev = thr
    if s >= 8:
        T["create"].append(b - a); T["start"].append(c - b); T["wait"].append(d - c); T["collect"].append(e - d)
t_loop = time.perf_counter()
for old in flying:
    old.wait(); old.close()
t_end = time.perf_counter()
n = steps - 8
print("depth", depth, "steady ms/step %.2f" % ((t_loop - t_begin) / n * 1e3), "with drain %.2f" % ((t_end - t_begin) / n * 1e3),
      {k: "%.2f" % (np.mean(v) * 1e3) for k, v in T.items()}, "create p90 %.2f" % (np.percentile(T["create"], 90) * 1e3),
      "wait p90 %.2f" % (np.percentile(T["wait"], 90) * 1e3))
