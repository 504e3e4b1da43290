#!/bin/bash
# SA_FLAG_INPUTS_IN_HOST_BLOCK: its tests, then the headline workload with two host threads (a rank's share of an 8-GPU run
# under a 16-CPU quota) with and without the flag, and with all threads
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/host_block
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_host_block.py tests/test_gpu_dplan.py -x -q > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; echo TESTS FAILED; exit 1; }
tail -n 2 $O/tests.log
run() { n=$1; shift; timeout -k 10 300 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 2 $O/$n.err | cut -c1-200; python -c "
import json,sys
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d['config'].get('long_run'))"; }
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run t2_block python bench.py --no-secondary --no-cpu-baseline --inputs host-block
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 SA_BLOCK_SPLIT=0 run t2_block_one_piece python bench.py --no-secondary --no-cpu-baseline --inputs host-block
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run t2_pageable python bench.py --no-secondary --no-cpu-baseline --inputs pageable
run all_block python bench.py --no-secondary --no-cpu-baseline --inputs host-block
run all_pageable python bench.py --no-secondary --no-cpu-baseline --inputs pageable
for d in 3 5; do   # (the default is four batches in flight)
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run t2_block_d$d python bench.py --no-secondary --no-cpu-baseline --inputs host-block --in-flight $d
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run t2_pageable_d$d python bench.py --no-secondary --no-cpu-baseline --inputs pageable --in-flight $d
run all_block_d$d python bench.py --no-secondary --no-cpu-baseline --inputs host-block --in-flight $d
run all_pageable_d$d python bench.py --no-secondary --no-cpu-baseline --inputs pageable --in-flight $d
done
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 SA_BENCH_DEBUG=1 SA_TRACE=1 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --inputs host-block --steps 6 --warmup 3 --long-steps 0 > $O/trace.json 2> $O/trace.err
tail -n 40 $O/trace.err | cut -c1-220
echo done
