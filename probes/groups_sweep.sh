#!/bin/bash
# result groups per pass (SA_GROUPS) against the kernels of one resident batch of a workload ($1): probes/groups_sweep.sh cpg 2 4 8 12 16
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
W=$1; shift
for g in "$@"; do
  if [ "$g" = default ]; then unset SA_GROUPS; else export SA_GROUPS=$g; fi
  python3 bench.py --workload $W --kernels-only --no-secondary --steps 4 --warmup 1 --no-cpu-baseline --full-record "" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W groups $g', '%.4g'%d['value'], '%.2f ms'%d['ms_per_step'], d['config']['kernel_ms'])"
done
