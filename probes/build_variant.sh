#!/bin/bash
# A library variant for probes/ab_variants.sh: probes/build_variant.sh <name> <-D flags...>  ->  probes/_variants/lib_<name>.so
# (the host objects of the regular build are reused; run __graft_entry__.build() first)
set -e
n=$1; shift
cd /root/repo/signalalign_amd
mkdir -p ../probes/_variants
T=$(mktemp -d)
F="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../include -Icsrc"
for f in sa_hip sa_ea sa_mea; do /opt/rocm/bin/hipcc $F "$@" -c csrc/$f.hip -o $T/$f.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../probes/_variants/lib_$n.so lib/sa_plan.o lib/sa_io.o lib/sa_hdpstate.o lib/sa_hdpgrid.o $T/sa_hip.o $T/sa_ea.o $T/sa_mea.o -lm -lpthread
rm -rf $T
echo built probes/_variants/lib_$n.so
