#!/bin/bash
# A/B of library variants on the expectation workload: probes/ab_expect.sh <name>...
for n in "$@"; do
  SA_LIBRARY=$PWD/probes/_variants/lib_$n.so python bench.py --workload expectations --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_$n.json || exit 1
  python - "$n" <<PY
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/ab_%s.json"%n).read().strip().splitlines()[-1]); print(n, "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"])
PY
done
