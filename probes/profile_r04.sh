#!/bin/bash
# Round 4: profile artefacts of ONE workload with one launch per stage (SA_GROUPS=1), so that the dominant kernel's average
# duration in the rocprofv3 kernel statistics IS the stage time bench.py reports (roofline.stage_ms / launches_per_step):
#   bench_kernels_only.json   the bench line of `bench.py --workload W --kernels-only --no-secondary` (plain run, fills the read cache)
#   kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command
#   traffic.json              FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, kernels only; gfx950 fetch doubling)
#   instr.json                instruction mix per kernel (two --pmc passes)
# Counter passes use --kernel-trace only (pool rule); the program itself stands behind `--`.
# usage: probes/profile_r04.sh <workload> [tag] [stats-only|full] [bench.py arguments...]
set -e
W=${1:-gaussian}
T=${2:-r04}
MODE=${3:-full}
shift 3 || true
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
O=gpurun_out/prof_${T}_$W
rm -rf $O && mkdir -p $O
if [ "$W" = expectations ] || [ "$W" = expectations_cpg ]; then
  CMD="bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline $EXTRA"; PASSES=6
else
  CMD="bench.py --workload $W --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline $EXTRA"; PASSES=4
fi
echo $PASSES > $O/passes
python3 $CMD > $O/bench_kernels_only.json 2> $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $CMD > $O/bench_under_profiler.json 2> $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
echo "[$W] kernel statistics done"
if [ "$MODE" = full ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $CMD > /dev/null 2> $O/pmc_$c.log
    echo "[$W] $c done"
  done
  python3 probes/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $PASSES > $O/traffic.json
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mix1 -- python3 $CMD > /dev/null 2> $O/pmc_mix1.log
  echo "[$W] mix1 done"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_mix2 -- python3 $CMD > /dev/null 2> $O/pmc_mix2.log
  python3 probes/pmc_summary.py $O/pmc_mix1 > $O/pmc_mix1.json
  python3 probes/pmc_summary.py $O/pmc_mix2 > $O/pmc_mix2.json
  python3 probes/instr_from_pmc.py $O/pmc_mix1.json $O/pmc_mix2.json $PASSES > $O/instr.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mix1 $O/pmc_mix2
fi
echo "profile of $W done"
