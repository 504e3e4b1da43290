#!/bin/bash
# Round 4 profiling session (one gpurun call): single-launch kernel statistics + counters for every bench workload.
# usage: probes/profile_session_r04.sh [tag]
T=${1:-r04}
set -e
bash probes/profile_r04.sh gaussian $T full
bash probes/profile_r04.sh realistic $T full
bash probes/profile_r04.sh hdp $T full
bash probes/profile_r04.sh expectations $T stats-only
bash probes/profile_r04.sh cpg $T full
bash probes/profile_r04.sh scaling $T stats-only
echo "session done"
