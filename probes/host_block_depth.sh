#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/host_block
mkdir -p $O
run() { n=$1; shift; timeout -k 10 400 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 2 $O/$n.err | cut -c1-200; python -c "
import json,sys
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d['config'].get('long_run'), {k: v.get('value') for k, v in (d['config'].get('secondary') or {}).items()})"; }
run default python bench.py --no-cpu-baseline
run default2 python bench.py --no-cpu-baseline
echo done
