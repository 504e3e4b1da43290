#!/bin/bash
# Profile artefacts of one bench workload (copied from gpurun_out/ into profiles/ afterwards): rocprofv3 kernel statistics of
# the default bench command, HBM traffic counters and instruction-mix counters over `bench.py --kernels-only`.  Counter passes
# use --kernel-trace only (pool rule) and the program itself behind `--`.  $1 = workload, $2 = tag (default r02)
set -e
W=${1:-gaussian}
T=${2:-r02}
shift 2 || true
EXTRA="$@"   # further bench.py arguments (the round-3 counters of cpg were collected at its former default of 2000 reads: --reads 2000)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_${T}_$W
rm -rf $O && mkdir -p $O
P=3   # kernel passes per counter run: --warmup 1 + --steps ... -> bench phase 1 runs min(warmup,3) + max(3,min(steps,10)) passes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $W --no-cpu-baseline $EXTRA > $O/bench_under_profiler.json 2> $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $O/pmc_$c.log
done
python3 probes/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 4 > $O/traffic.json
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mix1 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $O/pmc_mix1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_mix2 -- python3 bench.py --workload $W --kernels-only --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $O/pmc_mix2.log
python3 probes/pmc_summary.py $O/pmc_mix1 > $O/pmc_mix1.json
python3 probes/pmc_summary.py $O/pmc_mix2 > $O/pmc_mix2.json
python3 probes/instr_from_pmc.py $O/pmc_mix1.json $O/pmc_mix2.json 4 > $O/instr.json
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mix1 $O/pmc_mix2
echo "profile of $W done"
