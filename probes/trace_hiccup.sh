#!/bin/bash
# kernel + memory-copy trace of the pipelined headline: what the planner kernels of a slow sa_batch_create waited for
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_hic
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_hic -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trace_hic.log 2>&1
find gpurun_out/trace_hic -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/hic_kernels.csv
find gpurun_out/trace_hic -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/hic_copies.csv
rm -rf gpurun_out/trace_hic
grep -o '"ms_per_step": [0-9.]*' gpurun_out/trace_hic.log | head -1
