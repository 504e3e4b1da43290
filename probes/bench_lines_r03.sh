#!/bin/bash
# The bench lines of round 3 (copied from gpurun_out/ into profiles/bench_r03_*.json afterwards)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lines_r03
mkdir -p $O
run() { n=$1; shift; timeout -k 10 400 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 2 $O/$n.err | cut -c1-200; }
run gaussian python bench.py
run hdp python bench.py --workload hdp --no-cpu-baseline
run hdp_t001 python bench.py --workload hdp --threshold 0.01 --no-cpu-baseline
run cpg python bench.py --workload cpg --reads 2000 --no-cpu-baseline
run cpg10k python bench.py --workload cpg --steps 8 --warmup 2 --no-cpu-baseline
run realistic python bench.py --workload realistic --no-cpu-baseline
run scaling python bench.py --workload scaling --steps 8 --warmup 2 --no-cpu-baseline
run expectations python bench.py --workload expectations --steps 10 --warmup 2
run event_align python bench.py --workload event_align --steps 5 --warmup 2
run mea python bench.py --workload mea --steps 5 --warmup 2
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run gaussian_2threads python bench.py --no-secondary --no-cpu-baseline
SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run gaussian_2threads_dense_events python bench.py --no-secondary --no-cpu-baseline --event-stride 1
run gaussian_dense_events python bench.py --no-secondary --no-cpu-baseline --event-stride 1
SA_BENCH_BACKEND=gloo run gpus2_gloo_one_gpu python bench.py --gpus 2 --no-cpu-baseline --no-secondary
echo lines done
