#!/bin/bash
# kernel statistics of the PIPELINED headline run (fresh batches, several in flight, the bench's default launch shape), beside the
# single-launch statistics of probes/profile_r04.sh: profiles/<tag>_<workload>_kernel_stats_pipelined.csv
# usage: probes/profile_pipelined_r04.sh <workload> [tag] [bench arguments]
W=${1:-gaussian}; T=${2:-r04}; shift 2 || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
O=gpurun_out/prof_${T}_${W}_pipelined; rm -rf $O; mkdir -p $O
CMD="bench.py --workload $W --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --long-steps 0 $@"
python3 $CMD > $O/bench.json 2> $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $CMD > $O/bench_under_profiler.json 2> $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
echo "[$W] pipelined kernel statistics done"
