#!/bin/bash
# Round 4 profiling session: single-launch kernel statistics + counters for every bench workload, in two gpurun calls
# (a call is limited to 20 minutes).  usage: probes/profile_session_r04.sh [tag] [a|b]
T=${1:-r04}
PART=${2:-a}
set -e
if [ "$PART" = a ]; then
  bash probes/profile_r04.sh gaussian $T full
  bash probes/profile_r04.sh realistic $T full
  bash probes/profile_r04.sh hdp $T full
else
  bash probes/profile_r04.sh expectations $T full
  bash probes/profile_r04.sh expectations_cpg $T stats-only
  bash probes/profile_r04.sh cpg $T full
  bash probes/profile_r04.sh scaling $T stats-only
fi
echo "session $PART done"
