#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_expectations.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -x -q 2>&1 | tail -n 4
cp signalalign_amd/lib/libsignalalign_hip.so probes/_variants/lib_wmax_dpp.so
for w in gaussian realistic cpg hdp; do
  echo "== $w"
  timeout -k 10 600 bash probes/ab_variants.sh $w wred_shfl wmax_dpp wred_shfl wmax_dpp || exit 1
done
timeout -k 10 300 python bench.py --workload expectations --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/exp_wsum.json 2> gpurun_out/exp_wsum.err; python -c "
import json
d=json.loads(open('gpurun_out/exp_wsum.json').read().strip().splitlines()[-1]); print('expectations', d['value'], d['ms_per_step'], d['config'].get('kernel_ms'))"
