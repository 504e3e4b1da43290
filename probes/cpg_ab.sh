#!/bin/bash
# configs[2] kernels-only A/B of library variants (probes/_variants/lib_<name>.so): CPG_VARIANTS="base occ5 ..." [READS=2000]
cd "$GRAFT_REPO_ROOT"
export SA_SYNTH_CACHE=/tmp/sa_reads
for v in ${CPG_VARIANTS:-base}; do
  if [ $v = base ]; then unset SA_LIBRARY; else export SA_LIBRARY=$GRAFT_REPO_ROOT/probes/_variants/lib_$v.so; fi
  python3 bench.py --workload cpg --reads ${READS:-2000} --kernels-only --no-secondary --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/cpgab_$v.json 2> gpurun_out/cpgab_$v.err
  python3 -c "
import json; r=json.loads(open('gpurun_out/cpgab_$v.json').read().strip().splitlines()[-1]); print('$v', r['value'], r['config']['kernel_ms'])"
done
