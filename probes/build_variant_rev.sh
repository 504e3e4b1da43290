#!/bin/bash
# The library as of a git revision, as a probe variant: probes/build_variant_rev.sh <name> <rev> [-D flags]  ->  probes/_variants/lib_<name>.so
# (for A/B runs of two states of the kernels inside ONE gpurun call: boxes differ by several per cent)
set -e
n=$1; rev=$2; shift 2
cd /root/repo
T=$(mktemp -d)
git archive "$rev" signalalign_amd/csrc include | tar -x -C $T
mkdir -p probes/_variants
F="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$T/include -I$T/signalalign_amd/csrc"
/opt/rocm/bin/hipcc $F "$@" -c $T/signalalign_amd/csrc/sa_hip.hip -o $T/sa_hip.o
O=$(ls signalalign_amd/lib/*.o | grep -v "lib/sa_hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o probes/_variants/lib_$n.so $O $T/sa_hip.o -lm -lpthread
rm -rf $T
echo built probes/_variants/lib_$n.so
