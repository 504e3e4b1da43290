#!/bin/bash
# configs[2] at its 10 000 reads: one batch on the device at a time (134 GB of forward storage) against two batches in flight,
# each planned with half the forward-storage budget (two passes per batch)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { n=$1; shift; timeout -k 10 500 "$@" > gpurun_out/$n.json 2> gpurun_out/$n.err || { tail -n 5 gpurun_out/$n.err; echo FAILED $n; }; python - $n <<PY
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1]); c=d["config"]   # (the short line of round 6)
print(n, "%.4g"%d["value"], "%.1f ms"%d["ms_per_step"], "passes", c["forward_storage_passes"], "in flight", c["batches_in_flight"], "ko %.4g"%c["kernels_only_value"])
PY
}
run cpg10k_one python bench.py --workload cpg --steps 6 --warmup 2 --no-cpu-baseline
SA_F_BUDGET_CELLPATHS=2900000000 run cpg10k_two python bench.py --workload cpg --steps 6 --warmup 2 --no-cpu-baseline --in-flight 2
SA_F_BUDGET_CELLPATHS=1900000000 run cpg10k_three python bench.py --workload cpg --steps 6 --warmup 2 --no-cpu-baseline --in-flight 3
