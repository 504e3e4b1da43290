#!/bin/bash
# builds and runs the C streaming probe on the GPU box: bash probes/streaming_c/run.sh [reads per batch]
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
D=$(mktemp -d)
MODEL=$(python3 "$ROOT/probes/streaming_c/dump_jobs.py" "$D" "${1:-2000}")
gcc -O2 -std=gnu11 -I"$ROOT/include" -o "$D/stream" "$ROOT/probes/streaming_c/stream.c" -L"$ROOT/signalalign_amd/lib" -lsignalalign_hip -Wl,-rpath,"$ROOT/signalalign_amd/lib"
"$D/stream" "$MODEL" "$D/jobs_0.bin" "$D/jobs_1.bin" 16
rm -rf "$D"
