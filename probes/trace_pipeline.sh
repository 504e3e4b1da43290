#!/bin/bash
# kernel trace of the pipelined headline (fresh batches, several in flight): who overlaps whom
W=${1:-gaussian}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_pipe
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_pipe -- python3 bench.py --workload $W --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/trace_pipe.log 2>&1
find gpurun_out/trace_pipe -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/trace_pipe_trace.csv
rm -rf gpurun_out/trace_pipe
tail -1 gpurun_out/trace_pipe.log | cut -c1-300
