#!/bin/bash
# (SA_LIBRARY_BASE=<lib>: the library `base` and the env: labels run on, e.g. probes/_variants/lib_ringp.so)
# round 6: configs[2] at its 10 000 reads, kernels of one resident batch: the packed forward sweep with 1 / 2 / 4 shared waves, the
# unpacked one, and probe builds (probes/build_variant.sh).  Arguments: labels of the form env:VAR=V or lib:<variant> or base
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
for n in "$@"; do
  unset SA_RING_PACKED SA_RINGP_SHARED
  [ -n "$SA_LIBRARY_BASE" ] && export SA_LIBRARY=$SA_LIBRARY_BASE || unset SA_LIBRARY
  case "$n" in
    lib:*) export SA_LIBRARY=$PWD/probes/_variants/lib_${n#lib:}.so ;;
    env:*) export "${n#env:}" ;;
  esac
  t=$(echo "$n" | tr ':=' '__')
  python3 bench.py --workload cpg --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ringp_$t.json 2> gpurun_out/ringp_$t.err || { tail -3 gpurun_out/ringp_$t.err; continue; }
  python3 - "$n" "$t" <<PY
import json,sys
d=json.loads(open("gpurun_out/ringp_%s.json"%sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
