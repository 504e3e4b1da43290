#!/bin/bash
# k_fold tile width A/B (FOLD_TW 8 / 16 / 32): kernel statistics of the headline and the realistic workload, one launch per stage
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
for v in ${FOLD_VARIANTS:-base tw8 tw32}; do
  if [ $v = base ]; then unset SA_LIBRARY; else export SA_LIBRARY=$GRAFT_REPO_ROOT/probes/_variants/lib_$v.so; fi
  for w in gaussian realistic; do
    O=gpurun_out/foldab_${v}_$w; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $w --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/log
    f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if r['Name'].startswith('k_fold') or r['Name'].startswith('k_gather'): print('$v $w', r['Name'][:16], 'calls', r['Calls'], 'avg ms', float(r['AverageNs'])/1e6)"
    rm -rf $O/stats
  done
done
