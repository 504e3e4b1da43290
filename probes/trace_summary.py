"""Summary of a SA_TRACE=1 stderr log of bench.py: durations of sa_batch_create (and of its phases), of the runs (compute drained,
copies drained), pool misses with their cost.  usage: python probes/trace_summary.py log [skip_first_n_creates]"""
import re, sys
import numpy as np
L = open(sys.argv[1]).read().split("\n")
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
num = lambda l: float(re.search(r"at ([0-9.]+) ms", l).group(1))
creates, runs, copies, miss = [], [], [], []
phase = {}
n = 0
for l in L:
    if "create: done" in l:
        n += 1
        if n > skip: creates.append(num(l))
    for key in ("reads checked", "inputs packed", "kernels queued", "planned on the device", "inputs uploaded", "buffers allocated"):
        if key in l and n >= skip:
            phase.setdefault(key, []).append(num(l))
    if "compute stream drained" in l and n > skip: runs.append(num(l))
    if "copies drained" in l and n > skip: copies.append(num(l))
    if "pool: new" in l:
        ms = float(re.search(r": ([0-9.]+) ms", l).group(1))
        miss.append((n, ms, l.split("pool: ")[1]))
def q(v): return "n=%d median %.1f p90 %.1f max %.1f" % (len(v), np.median(v), np.percentile(v, 90), max(v)) if v else "-"
print("create:", q(creates))
for k, v in phase.items(): print("  %-22s %s" % (k, q(v)))
print("run, compute drained:", q(runs))
print("run, copies drained: ", q(copies))
late = [m for m in miss if m[0] > skip]
print("pool misses: %d in all, %d after the first %d creates:" % (len(miss), len(late), skip), [(m[0], round(m[1], 1), m[2][:40]) for m in late][:8])
