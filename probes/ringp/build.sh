#!/bin/bash
# Builds the library with the packed forward ring sweep of round 6 (probes/ringp/sa_ringp.inc: four regions with several paths per
# cell to a workgroup) as a probe variant: probes/_variants/lib_ringp.so.  The product does not contain this kernel: measured on
# configs[2] it gives the same bytes and no gain (probes/ringp/README.md).  The patch adds the include and routes the launch of the
# classes with rows of at most 128 cell-paths; SA_RING_PACKED=0 falls back to k_fwd_ring inside the variant.
set -e
cd /root/repo
T=$(mktemp -d)
cp -r signalalign_amd/csrc include $T/
cp probes/ringp/sa_ringp.inc $T/csrc/
python3 - $T/csrc/sa_hip.hip <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace('#include "sa_ring.inc"\n', '#include "sa_ring.inc"\n#include "sa_ringp.inc"\n', 1)
old = "                    launch_fwd_ring(P, b->d_ids + C.ids_rr[cl], C.nrr[cl], lanes[n_lanes > 1 ? which : 0], 64 * ((cl & 7) + 1), cl >= 8);"
new = ("                    if ((cl == 8 || cl == 9) && ring_packed_on())\n"
       "                        launch_fwd_ringp(P, b->d_ids + C.ids_rr[cl], C.nrr[cl], lanes[n_lanes > 1 ? which : 0], 64 * ((cl & 7) + 1));\n"
       "                    else\n    " + old)
assert old in s
open(p, 'w').write(s.replace(old, new))
PY
mkdir -p probes/_variants
F="-O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$T/include -I$T/csrc"
/opt/rocm/bin/hipcc $F "$@" -c $T/csrc/sa_hip.hip -o $T/sa_hip.o
O=$(ls signalalign_amd/lib/*.o | grep -v "lib/sa_hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o probes/_variants/lib_ringp.so $O $T/sa_hip.o -lm -lpthread
rm -rf $T
echo built probes/_variants/lib_ringp.so
