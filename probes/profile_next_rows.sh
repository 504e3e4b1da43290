#!/bin/bash
# rocprofv3 kernel statistics of the two "next" rows and of the headline bench, same commands
# as bench.py runs them.  Output under gpurun_out/; the summaries worth keeping are copied to profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ea -- python3 bench.py --workload event_align --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_ea.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mea -- python3 bench.py --workload mea --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_mea.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_main -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_main.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real -- python3 bench.py --workload realistic --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_real.log 2>&1
python3 bench.py > gpurun_out/bench_main.json 2> gpurun_out/bench_main.err
python3 bench.py --workload event_align > gpurun_out/bench_ea.json 2> gpurun_out/bench_ea.err
python3 bench.py --workload mea > gpurun_out/bench_mea.json 2> gpurun_out/bench_mea.err
python3 bench.py --workload realistic > gpurun_out/bench_real.json 2> gpurun_out/bench_real.err
echo done
