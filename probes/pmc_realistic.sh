#!/bin/bash
# instruction mix and wait fractions of the sweeps on the realistic-anchor workload (wide bands): what does a wave of
# k_bwd_fast_wide wait for?  Two counter passes (kernel trace only, as the pool requires), summed per kernel.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_real1 -- python3 bench.py --workload realistic --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_real1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_real2 -- python3 bench.py --workload realistic --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_real2.log 2>&1
python3 probes/pmc_summary.py gpurun_out/pmc_real1 > gpurun_out/pmc_real1.json
python3 probes/pmc_summary.py gpurun_out/pmc_real2 > gpurun_out/pmc_real2.json
echo done
