#!/bin/bash
# config.scaling_job (BASELINE configs[4] as a strong-scaling job) on one GPU: the job alone at several slice sizes, then the
# N > 1 path rehearsed with 2 and 4 gloo ranks sharing the one card (lines copied to profiles/bench_r05_scaling_job_*.json)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lines_r05
mkdir -p $O
run() { n=$1; shift; timeout -k 10 600 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 2 $O/$n.err | cut -c1-200; }
for sl in ${SLICES:-2000 4000 12500}; do
  run scaling_job_slice$sl python3 bench.py --workload scaling_job --job-slice $sl
done
if [ "${GLOO:-1}" = 1 ]; then
  SA_BENCH_BACKEND=gloo run scaling_job_gpus2_gloo_one_gpu python3 bench.py --gpus 2 --workload scaling_job
  SA_BENCH_BACKEND=gloo run scaling_job_gpus4_gloo_one_gpu python3 bench.py --gpus 4 --workload scaling_job
fi
for f in $O/scaling_job_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))["config"]
    print(sys.argv[1].split("/")[-1], "value %.3e wall %.2f s depth %s passes %s slices %s per-rank %s" % (
        j["value"], j["wall_s"], j["batches_in_flight"], j["forward_storage_passes_per_slice"], j["slices_rank0"],
        [round(r["value"] / 1e9, 1) for r in j["per_rank"]]))
except Exception as ex:
    print(sys.argv[1], "unreadable:", ex)
PY
done
echo lines done
