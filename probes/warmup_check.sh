#!/bin/bash
# the headline line must not depend on the warm-up count the driver happens to pass (the allocators are primed before it)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in 0 1 5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --warmup $w --steps 20 > gpurun_out/wu_$w.json 2> gpurun_out/wu_$w.err || { tail -n 5 gpurun_out/wu_$w.err; exit 1; }
  python - $w <<PY
import json,sys
w=sys.argv[1]
d=json.loads(open("gpurun_out/wu_%s.json"%w).read().strip().splitlines()[-1]); print("warmup", w, "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"]["allocator_priming_batches_before_warmup"])
PY
done
