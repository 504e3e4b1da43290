#!/bin/bash
# round 6: config.scaling_job (BASELINE configs[4] as a strong-scaling job, five repetitions inside one launch, equal slices) on one
# GPU: the job alone, then the N > 1 path rehearsed with 2 and 4 gloo ranks sharing the one card (lines -> profiles/bench_r06_scaling_job_*.json)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lines_r06
mkdir -p $O
run() { n=$1; shift; timeout -k 10 500 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 2 $O/$n.err | cut -c1-200; }
run scaling_job_n1 python3 bench.py --workload scaling_job --full-record ''
SA_BENCH_BACKEND=gloo run scaling_job_gpus2_gloo_one_gpu python3 bench.py --gpus 2 --workload scaling_job --full-record ''
SA_BENCH_BACKEND=gloo run scaling_job_gpus4_gloo_one_gpu_40000_reads python3 bench.py --gpus 4 --workload scaling_job --job-reads 40000 --full-record ''
for f in $O/scaling_job_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["config"]   # (gloo prints its connection lines to stdout first)
    print(sys.argv[1].split("/")[-1], "value %.3e wall median %.3f s (min %.3f max %.3f, spread %.1f %%) depth %s slices %s %s idle at the barrier %s" % (
        j["value"], j["wall_s"], j["wall_s_min"], j["wall_s_max"], 100 * j["wall_spread"], j["batches_in_flight"], j["slices_rank0"],
        j["slice_sizes_rank0"], [round(r["idle_at_barrier_s"], 3) for r in j["per_rank"]]))
except Exception as ex:
    print(sys.argv[1], "unreadable:", ex)
PY
done
echo lines done
