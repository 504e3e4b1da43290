#!/bin/bash
# Which leg of the default run slows the scaling_slice leg behind it?  usage: probes/leg_context.sh <leg>...
for l in "$@"; do
  SA_TRACE=1 timeout -k 10 500 python bench.py --legs $l,scaling_slice --no-scaling-job --no-cpu-baseline --steps 4 --warmup 2 --full-record gpurun_out/legctx_$l.full.json > gpurun_out/legctx_$l.json 2> gpurun_out/legctx_$l.err || exit 1
  python3 - "$l" <<PY
import json,sys
l=sys.argv[1]
d=json.load(open("gpurun_out/legctx_%s.full.json"%l))   # (the complete record: the printed line is its short form since round 6)
s=d["config"]["secondary"]
print(l, "->", {k:(round(v["ms_per_step"],1) if "ms_per_step" in v else v) for k,v in s.items() if k in (l,"scaling_slice")}, flush=True)
PY
done
