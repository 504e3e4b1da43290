#!/bin/bash
# kernel trace of the realistic-anchor workload (kernels only): per-launch durations of the strip kernels
W=${1:-realistic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_strip
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_strip -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/trace_strip.log 2>&1
find gpurun_out/trace_strip -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/trace_strip_stats.csv
find gpurun_out/trace_strip -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/trace_strip_trace.csv
rm -rf gpurun_out/trace_strip
