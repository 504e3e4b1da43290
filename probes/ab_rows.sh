#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp signalalign_amd/lib/libsignalalign_hip.so probes/_variants/lib_rows48.so
for w in gaussian realistic cpg hdp; do
  echo "== $w"
  timeout -k 10 600 bash probes/ab_variants.sh $w rows32 rows48 rows32 rows48 || exit 1
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -n 3
