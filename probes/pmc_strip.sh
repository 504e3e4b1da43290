#!/bin/bash
# two counter passes over the realistic-anchor workload (kernels only): instruction mix and wait fractions per kernel
# $1 = workload (default realistic), $2 = tag
W=${1:-realistic}
T=${2:-strip}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${T}1 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${T}1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_${T}2 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${T}2.log 2>&1
python3 probes/pmc_summary.py gpurun_out/pmc_${T}1 > gpurun_out/pmc_${T}1.json
python3 probes/pmc_summary.py gpurun_out/pmc_${T}2 > gpurun_out/pmc_${T}2.json
rm -rf gpurun_out/pmc_${T}1 gpurun_out/pmc_${T}2
