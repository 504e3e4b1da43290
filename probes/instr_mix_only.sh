#!/bin/bash
# Instruction-mix counters of one bench workload (two rocprofv3 --pmc passes over `bench.py --kernels-only`), written to
# gpurun_out/mix_<workload>/instr.json.  $1 = workload
set -e
W=${1:-cpg}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mix_$W
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mix1 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mix1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_mix2 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mix2.log
python3 probes/pmc_summary.py $O/pmc_mix1 > $O/pmc_mix1.json
python3 probes/pmc_summary.py $O/pmc_mix2 > $O/pmc_mix2.json
python3 probes/instr_from_pmc.py $O/pmc_mix1.json $O/pmc_mix2.json 4 > $O/instr.json
rm -rf $O/pmc_mix1 $O/pmc_mix2
echo "mix of $W done"
