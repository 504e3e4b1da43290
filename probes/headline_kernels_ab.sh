#!/bin/bash
# kernels of one resident batch of a workload ($1) for the regular build (base) and probe builds (lib:<variant>)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
W=$1; shift
for n in "$@"; do
  unset SA_LIBRARY
  case "$n" in lib:*) export SA_LIBRARY=$PWD/probes/_variants/lib_${n#lib:}.so ;; esac
  t=$(echo "$n" | tr ':=' '__')
  python3 bench.py --workload $W --kernels-only --no-secondary --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/hk_${W}_$t.json 2> gpurun_out/hk_${W}_$t.err || { tail -3 gpurun_out/hk_${W}_$t.err; continue; }
  python3 - "$n" "$W" "$t" <<PY
import json,sys
d=json.loads(open("gpurun_out/hk_%s_%s.json"%(sys.argv[2],sys.argv[3])).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
