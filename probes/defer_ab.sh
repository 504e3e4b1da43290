#!/bin/bash
# round 6: the headline's pipelined loop with sa_batch_create (one piece, the default) against sa_batch_create_deferred + start (the
# second half of the creation -- waiting for the device planner, launch lists, buffers -- on the batch's runner thread), alternating
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for d in 0 1; do
  if [ $d = 1 ]; then export SA_BENCH_DEFER=1; else unset SA_BENCH_DEFER; fi
  timeout -k 10 200 python3 bench.py --no-secondary --no-cpu-baseline --no-scaling-job --full-record "" --steps 40 --warmup 8 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['step_ms']
print('defer=$d', '%.4g'%d['value'], 'ms/step %.2f'%d['ms_per_step'], 'p10/p50/p90/max', s['p10'], s['p50'], s['p90'], s['max'], 'create', s['median_create_ms'], 'wait', s['median_wait_ms'])"
done; done
