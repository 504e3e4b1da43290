#!/bin/bash
# A library variant from a sed-patched copy of the kernel sources: probes/build_variant_sed.sh <name> <file under csrc> <sed expression> [...]
# (A/B experiments without build-time macros in the product sources; pairs of <file> <expression> may repeat)
set -e
n=$1; shift
cd /root/repo
T=$(mktemp -d)
cp -r signalalign_amd/csrc include $T/
while [ $# -ge 2 ]; do
  sed -i -E "$2" $T/csrc/$1
  shift 2
done
mkdir -p probes/_variants
F="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$T/include -I$T/csrc"
/opt/rocm/bin/hipcc $F -c $T/csrc/sa_hip.hip -o $T/sa_hip.o
O=$(ls signalalign_amd/lib/*.o | grep -v "lib/sa_hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o probes/_variants/lib_$n.so $O $T/sa_hip.o -lm -lpthread
diff <(cd signalalign_amd/csrc && cat sa_ring.inc sa_fast.inc sa_hip.hip sa_strip.inc) <(cd $T/csrc && cat sa_ring.inc sa_fast.inc sa_hip.hip sa_strip.inc) | head -12
rm -rf $T
echo built probes/_variants/lib_$n.so
