#!/bin/bash
# kernels-only rate of the headline workload against the number of reads per batch (waves per launch)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for r in 500 1000 2000 4000 8000; do
  timeout -k 10 300 python bench.py --kernels-only --no-cpu-baseline --reads $r --steps 6 --warmup 2 > gpurun_out/sweep_$r.json 2> gpurun_out/sweep_$r.err || { tail -n 5 gpurun_out/sweep_$r.err; exit 1; }
  python - $r <<PY
import json,sys
r=sys.argv[1]
d=json.loads(open("gpurun_out/sweep_%s.json"%r).read().strip().splitlines()[-1]); k=d["config"]["kernel_ms"]
print(r, "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "per 1000 reads: fwd %.2f bwd %.2f"%(k["forward"]*1000/int(r), k["backward_posterior"]*1000/int(r)))
PY
done
