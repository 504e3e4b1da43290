#!/bin/bash
# k_emit_hdp's average duration per library variant (probes/build_variant.sh), kernels of the resident configs[3] batch.
# usage: probes/emit_variants.sh <name>...      ("base" = the regular library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_GROUPS=1 SA_SYNTH_CACHE=/tmp/sa_reads
python3 bench.py --workload hdp --kernels-only --no-secondary --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for n in "$@"; do
  if [ "$n" = base ]; then unset SA_LIBRARY; else export SA_LIBRARY=$PWD/probes/_variants/lib_$n.so; fi
  O=gpurun_out/emitv_$n
  rm -rf $O && mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload hdp --kernels-only --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/log || exit 1
  f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
  echo "== $n"; grep -E "k_emit_hdp|k_fwd_fast_hdp|k_bwd_fast_hdp" $f | awk -F'",' '{split($2,a,","); printf "%s avg %.3f ms\n", substr($1,2,18), a[3]/1e6}'
  rm -rf $O/stats
done
