#!/bin/bash
# What the second predecessor's / successor's fold costs the ring kernels on configs[2] (probe builds: wrong results, timing only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
for n in "$@"; do
  if [ "$n" = base ]; then unset SA_LIBRARY; else export SA_LIBRARY=$PWD/probes/_variants/lib_$n.so; fi
  python3 bench.py --workload cpg --reads 4000 --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ringp_$n.json 2> gpurun_out/ringp_$n.err || { tail -3 gpurun_out/ringp_$n.err; continue; }
  python3 - "$n" <<PY
import json,sys
d=json.loads(open("gpurun_out/ringp_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
