#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of `bench.py --workload $1 --kernels-only` for the regular build
# (base) and variant libraries (lib:<name>), alternating inside one gpurun call; $2 = regular expression of the kernels to print
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads SA_GROUPS=1
W=$1; PAT=$2; shift 2
for n in "$@"; do
  unset SA_LIBRARY
  case "$n" in lib:*) export SA_LIBRARY=$PWD/probes/_variants/lib_${n#lib:}.so ;; esac
  O=gpurun_out/kab_$W_$(echo "$n" | tr ':=' '__')
  rm -rf $O && mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $W --kernels-only --no-secondary --steps 5 --warmup 1 --no-cpu-baseline --full-record "" > $O/line.json 2> $O/err.log
  f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
  echo "== $n"
  python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("   %-44s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/stats
done
