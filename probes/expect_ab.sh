#!/bin/bash
# the expectation pass with several paths per cell (bench.py --workload expectations_cpg) for the regular build (base) and variants
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
for n in "$@"; do
  unset SA_LIBRARY
  case "$n" in lib:*) export SA_LIBRARY=$PWD/probes/_variants/lib_${n#lib:}.so ;; esac
  t=$(echo "$n" | tr ':=' '__')
  python3 bench.py --workload expectations_cpg --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/xab_$t.json 2> gpurun_out/xab_$t.err || { tail -3 gpurun_out/xab_$t.err; continue; }
  python3 - "$n" "$t" <<PY
import json,sys
d=json.loads(open("gpurun_out/xab_%s.json"%sys.argv[2]).read().strip().splitlines()[-1]); print("expectations_cpg", sys.argv[1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"].get("mean_match_to_match_expectation"), d["config"].get("mean_log_likelihood"))
PY
done
