#!/bin/bash
# headline kernels A/B of library variants: kernel statistics (one launch per stage) of gaussian and hdp; HL_VARIANTS="base occ6 ..."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
for v in ${HL_VARIANTS:-base}; do
  if [ $v = base ]; then unset SA_LIBRARY; else export SA_LIBRARY=$GRAFT_REPO_ROOT/probes/_variants/lib_$v.so; fi
  for w in ${HL_WORKLOADS:-gaussian hdp}; do
    O=gpurun_out/hlab_${v}_$w; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $w --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/log
    f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
    python3 -c "
import csv,sys,json
for r in csv.DictReader(open('$f')):
    if r['Name'].startswith(('k_bwd','k_fwd','void k_bwd','void k_fwd')): print('$v $w', r['Name'][:22], 'avg ms', round(float(r['AverageNs'])/1e6,3))
print('$v $w value', json.loads(open('$O/b.json').read().strip().splitlines()[-1])['value'])"
    rm -rf $O/stats
  done
done
