#!/bin/bash
# rocprofv3 kernel statistics of `bench.py --workload $1 --kernels-only` (one planned, resident batch run repeatedly): the
# per-kernel split of a batch.  $1 = workload, $2 = tag, further arguments go to bench.py.
set -e
W=${1:-hdp}
T=${2:-r03}
shift 2 || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stats_${T}_$W
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $W --kernels-only --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/bench_under_profiler.json 2> $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
cat $O/kernel_stats.csv | cut -c1-200 | head -20
