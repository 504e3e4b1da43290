#!/bin/bash
# usage: probes/hiccup_ab.sh "<env assignments / bench flags variant 1>" "<variant 2>" ...  -- each variant: "ENV=.. ENV=.. -- flags"
for rep in 1 2 3; do
  for v in "$@"; do
    envs="${v%%--*}"; flags="${v#*--}"
    env $envs SA_BENCH_DEBUG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $flags > gpurun_out/box.json 2> gpurun_out/dbg.err
    python - "$v" <<'PY'
import json,re,sys
d=json.loads(open("gpurun_out/box.json").read().strip().splitlines()[-1])
L=[l for l in open("gpurun_out/dbg.err") if "[bench] step" in l][-20:]
cr=[float(re.search(r"create ([0-9.]+)",l).group(1)) for l in L]; st=[float(re.search(r"step ([0-9.]+) ms",l).group(1)) for l in L]
print("%-40s %.2f ms; create mean %.1f max %.1f; steps>20ms: %d" % (sys.argv[1], d["ms_per_step"], sum(cr)/len(cr), max(cr), sum(1 for x in st if x>20)))
PY
  done
done
