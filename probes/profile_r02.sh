#!/bin/bash
# Round-2 profile artefacts (copied from gpurun_out/ into profiles/ afterwards): kernel statistics, HBM traffic counters and
# instruction-mix counters of the bench workloads.  Counter passes use --kernel-trace only (pool rule) and the program itself
# behind `--` (python3 bench.py ...).  $1 = workload (gaussian | realistic | cpg), $2 = extra bench flags (optional)
set -e
W=${1:-gaussian}
X=${2:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02_$W
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $W $X --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --workload $W $X --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --workload $W $X --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.log
python3 probes/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write 4 > $O/traffic.json
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mix1 -- python3 bench.py --workload $W $X --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mix1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_mix2 -- python3 bench.py --workload $W $X --kernels-only --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mix2.log
python3 probes/pmc_summary.py $O/pmc_mix1 > $O/pmc_mix1.json
python3 probes/pmc_summary.py $O/pmc_mix2 > $O/pmc_mix2.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mix1 $O/pmc_mix2
python3 bench.py --workload $W $X $( [ "$W" = gaussian ] || echo --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
echo "profile of $W done"
