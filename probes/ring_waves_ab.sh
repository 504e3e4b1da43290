#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in 0 1 2; do
  SA_RING_WAVES=$v timeout -k 10 300 python bench.py --workload cpg --reads ${READS:-2000} --kernels-only --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/rw_$v.json 2> gpurun_out/rw_$v.err || { tail -n 5 gpurun_out/rw_$v.err; exit 1; }
  python - $v <<PY
import json,sys
v=sys.argv[1]
d=json.loads(open("gpurun_out/rw_%s.json"%v).read().strip().splitlines()[-1]); print("SA_RING_WAVES="+v, "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
