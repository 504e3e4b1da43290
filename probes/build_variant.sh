#!/bin/bash
# A library variant for probes/ab_variants.sh: probes/build_variant.sh <name> <-D flags...>  ->  probes/_variants/lib_<name>.so
# (sa_hip.hip is compiled with the flags; every other object of the regular build is reused: run __graft_entry__.build() first)
set -e
n=$1; shift
cd /root/repo/signalalign_amd
mkdir -p ../probes/_variants
T=$(mktemp -d)
F="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../include -Icsrc"
/opt/rocm/bin/hipcc $F "$@" -c csrc/sa_hip.hip -o $T/sa_hip.o
O=$(ls lib/*.o | grep -v "lib/sa_hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../probes/_variants/lib_$n.so $O $T/sa_hip.o -lm -lpthread
rm -rf $T
echo built probes/_variants/lib_$n.so
