#!/bin/bash
# Register counts and spills of every kernel in sa_hip.hip (no GPU needed): probes/kernel_regs.sh [pattern]
set -e
T=$(mktemp -d) && cd "$T"
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I/root/repo/include \
    -I/root/repo/signalalign_amd/csrc ${SA_DEFS} -c /root/repo/signalalign_amd/csrc/sa_hip.hip -o x.o -save-temps 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|name|vgpr_spill_count|sgpr_spill_count):" sa_hip-hip-amdgcn-amd-amdhsa-gfx950.s |
    paste - - - - - | sed 's/ \+/ /g; s/\t/ /g' |
    awk '{print $2, "sgpr", $4, "sspill", $6, "vgpr", $8, "vspill", $10}' | c++filt | sed 's/(.*)//' | { grep -E "${1:-.}" || true; }
cp sa_hip-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/sa_hip_last.s
rm -rf "$T"
