#!/bin/bash
# How busy the GPU is during the pipelined headline loop: union of the kernel intervals of a kernel trace over the last second of
# the run (the 200-step long run), and the same per stream-overlap depth.  usage: probes/pipeline_busy.sh [bench arguments]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
O=gpurun_out/pipeline_busy; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --secondary-budget-s 0 --no-cpu-baseline --steps 20 --warmup 5 --long-steps 200 "$@" > $O/bench.json 2> $O/log
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:24]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t_end = max(e for _, e, _ in rows)
win = [(s, e, n) for s, e, n in rows if s >= t_end - 1_000_000_000]       # the last second: inside the long run
t0, t1 = min(s for s, _, _ in win), max(e for _, e, _ in win)
ev = sorted([(s, 1) for s, _, _ in win] + [(e, -1) for _, e, _ in win])
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = t1 - t0
print("window %.1f ms; kernels running at once -> share of the time: %s" % (tot / 1e6, {k: round(v / tot, 3) for k, v in sorted(hist.items())}))
by = {}
for s, e, n in win: by[n] = by.get(n, 0) + (e - s)
print("kernel time / window:", {k: round(v / tot, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]})
PY
rm -rf $O/trace
