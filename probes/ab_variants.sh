#!/bin/bash
# A/B of library variants (probes/_variants/lib_<name>.so, built with extra -D flags) on one workload, kernels only.
# usage: probes/ab_variants.sh <workload> <name>...
w=$1; shift
for n in "$@"; do
  SA_LIBRARY=$PWD/probes/_variants/lib_$n.so python bench.py --workload $w --kernels-only --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_$n.json || exit 1
  python - "$n" <<PY
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/ab_%s.json"%n).read().strip().splitlines()[-1]); print(n, "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"].get("kernel_ms"))
PY
done
