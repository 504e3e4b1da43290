#!/bin/bash
# Round 6 profiling session (final kernels): single-launch kernel statistics for every bench workload, counters for the workloads
# whose kernels changed this round or are new (cpg: k_bwd_ring; hdp: the emission kernels' conversion; hdp_dense: new, with the L2
# hit ratio of its table reads) and for the headline.  usage: probes/profile_session_r06.sh [a|b|c]   (a gpurun call is 20 minutes)
T=r06
PART=${1:-a}
set -e
case "$PART" in
a)
  bash probes/profile_r04.sh gaussian $T full
  bash probes/profile_r04.sh hdp $T full
  bash probes/profile_r04.sh realistic $T stats-only
  ;;
b)
  bash probes/profile_r04.sh cpg $T full
  bash probes/profile_r04.sh expectations_cpg $T stats-only
  bash probes/profile_r04.sh scaling $T stats-only
  ;;
c)
  bash probes/profile_r04.sh hdp_dense $T full
  bash probes/profile_r04.sh hdp_realistic $T stats-only
  bash probes/profile_r04.sh expectations $T stats-only
  # the table reads of the dense model: L2 hits and misses per kernel (one pass)
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
  for W in hdp_dense hdp; do
    O=gpurun_out/prof_${T}_$W
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 bench.py --workload $W --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline --full-record "" > /dev/null 2> $O/pmc_tcc.log || echo "TCC pass failed for $W"
    python3 probes/pmc_summary.py $O/pmc_tcc > $O/pmc_tcc.json || true
    rm -rf $O/pmc_tcc
  done
  ;;
esac
echo "session $PART done"
