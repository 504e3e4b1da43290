#!/bin/bash
# The bench lines of round 4 (copied from gpurun_out/lines_r04/ into profiles/bench_r04_*.json afterwards: probes/collect_lines_r04.py)
# part a: the driver's line and the workloads beside it; part b: the single-launch kernels-only lines (after collect_profiles.py, so
# that their counter fields come from this round's passes)
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
O=gpurun_out/lines_r04
mkdir -p $O
run() { n=$1; shift; timeout -k 10 500 "$@" > $O/$n.json 2> $O/$n.err || echo "FAILED $n"; tail -n 1 $O/$n.err | cut -c1-160; }
if [ "${1:-a}" = a ]; then
  run default python3 bench.py --gpus 1 --steps 20 --warmup 5
  run hdp_cpg python3 bench.py --workload hdp_cpg --threshold 0.1 --no-secondary
  run hdp_realistic python3 bench.py --workload hdp_realistic --threshold 0.1 --no-secondary
  run cpg_2000 python3 bench.py --workload cpg --reads 2000 --no-secondary --no-cpu-baseline
  run expectations_cpg python3 bench.py --workload expectations_cpg --reads 2000 --steps 5 --warmup 1
  run event_align python3 bench.py --workload event_align --steps 5 --warmup 2
  run mea python3 bench.py --workload mea --steps 5 --warmup 2
  SA_HOST_THREADS=2 SA_PLAN_THREADS=2 run gaussian_2threads python3 bench.py --no-secondary --no-cpu-baseline
  SA_BENCH_BACKEND=gloo run gpus2_gloo_one_gpu python3 bench.py --gpus 2 --no-cpu-baseline --no-secondary
else
  export SA_GROUPS=1
  for w in gaussian realistic hdp cpg scaling; do
    run ${w}_kernels_only_groups1 python3 bench.py --workload $w --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline
  done
fi
echo lines done
