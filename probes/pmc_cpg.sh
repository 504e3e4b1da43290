#!/bin/bash
# instruction mix and wait fractions of the memory-resident kernels on the CpG workload (configs[2])
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_cpg1 -- python3 bench.py --workload cpg --reads 2000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_cpg1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_cpg2 -- python3 bench.py --workload cpg --reads 2000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_cpg2.log 2>&1
echo done
