#!/bin/bash
# LDS-array cycles and bank-conflict cycles of one bench workload's kernels (rocprofv3 --pmc over `bench.py --kernels-only`).
# $1 = workload -> gpurun_out/lds_<workload>/lds.json
set -e
W=${1:-gaussian}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lds_$W
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc.log || { tail -n 20 $O/pmc.log; exit 1; }
python3 probes/pmc_summary.py $O/pmc > $O/lds.json
rm -rf $O/pmc
python3 - <<PY
import json
d=json.load(open("$O/lds.json"))
for k,c in sorted(d.items(), key=lambda kv:-kv[1].get("SQ_LDS_IDX_ACTIVE",0))[:8]:
    print(k, {n:(round(v,4) if isinstance(v,float) and v<10 else int(v)) for n,v in c.items() if "LDS" in n or n in ("SQ_BUSY_CYCLES","SQ_WAVES","launches")})
PY
