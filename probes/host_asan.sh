#!/bin/bash
# The host sources (planner, loaders, anchors, parameter estimation, the .nhdp state, the expectations objects, the HDP rebuild's host
# side: sa_plan.c, sa_io.c, sa_hdpstate.c, sa_hmm.c, sa_hdpgibbs.c) built with
# gcc under AddressSanitizer + UBSan and driven by the host test-suite.  Every exported entry point that lives in a HIP source is a
# stub that returns SA_ENODEVICE (generated from the library's own export list): this build exists to check host memory safety
# only (GPU sanitizers are not available on the pool).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/sa_host_asan
mkdir -p $OUT
SRC="$ROOT/signalalign_amd/csrc/sa_plan.c $ROOT/signalalign_amd/csrc/sa_io.c $ROOT/signalalign_amd/csrc/sa_hdpstate.c $ROOT/signalalign_amd/csrc/sa_hmm.c $ROOT/signalalign_amd/csrc/sa_hdpgibbs.c"
FLAGS="-O1 -g -std=c11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -I$ROOT/include -I$ROOT/signalalign_amd/csrc"
# which exports the host sources do not define
gcc $FLAGS -shared -o $OUT/host_only.so $SRC -lm -lpthread
python3 - "$ROOT" $OUT <<'PY'
import re, subprocess, sys
root, out = sys.argv[1], sys.argv[2]
src = open(root + "/signalalign_amd/_capi.py").read()
exports = re.findall(r'"(sa_[a-z0-9_]+)"', src[src.index("EXPORTS"):src.index("]", src.index("EXPORTS"))])
defined = set(l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", out + "/host_only.so"], capture_output=True,
                                                    text=True).stdout.splitlines() if l.strip())
# ... and what the host sources call inside the HIP sources without exporting it (sa_hdp_sampler_*)
undefined = set(l.split()[-1] for l in subprocess.run(["nm", "-D", "--undefined-only", out + "/host_only.so"], capture_output=True,
                                                      text=True).stdout.splitlines() if l.strip() and l.split()[-1].startswith("sa_"))
with open(out + "/stubs.c", "w") as f:
    f.write("/* generated: entry points that live in HIP sources (no prototypes on purpose: they only return SA_ENODEVICE) */\n")
    for name in sorted((set(exports) | undefined) - defined):
        if name in ("sa_event_align_release", "sa_mea_release", "sa_batch_destroy", "sa_host_free", "sa_free", "sa_hdp_sampler_close"):
            f.write("void %s() {}\n" % name)
        elif name == "sa_pool_release":
            f.write("void sa_plan_pool_release(void); void sa_pool_release() { sa_plan_pool_release(); }\n")
        elif name in ("sa_hdp_sampler_open", "sa_hdp_sampler_add", "sa_hdp_sampler_finish"):
            # the Gibbs sweeps run on the host: with a sampler that accepts everything they can be driven here (below)
            f.write("int %s() { return 0; }\n" % name)
        elif name == "sa_device_count":
            f.write("int sa_device_count() { return 0; }\n")
        elif name == "sa_host_alloc":
            f.write("void *sa_host_alloc() { return 0; }\n")
        elif name in ("sa_mea_printed_posterior",):
            f.write("double %s(long long p) { return (double) p; }\n" % name)
        else:
            f.write("int %s() { return -3; }\n" % name)
print("stubs:", len(set(exports) - defined))
PY
grep -q "define SA_ENODEVICE (-3)" "$ROOT/include/signalalign_hip.h" || { echo "SA_ENODEVICE is not -3: update the stub generator"; exit 1; }
gcc $FLAGS -Wno-implicit-function-declaration -shared -o $OUT/libsignalalign_hip.so $SRC $OUT/stubs.c -lm -lpthread
cd "$ROOT"
SA_SAMPLER_STUB=1 SA_LIBRARY=$OUT/libsignalalign_hip.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_plan.py tests/test_host_hdp_state.py tests/test_host_hmm.py tests/test_host_hdp_gibbs.py -x -q -p no:cacheprovider -k "not scalings_by_method_of_moments" "$@"   # that one lives in a HIP source

# the Gibbs sweeps of the HDP rebuild (sa_hdpgibbs.c: an index tree that every iteration rewires) under the sanitizers, with the
# GPU sampler stubbed out: the reference's test HDP and its data, a flat and a multiset NanoporeHDP on the reference's alignment table
SA_LIBRARY=$OUT/libsignalalign_hip.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 - <<'PY'
import gzip, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import signalalign_amd as sa
from signalalign_amd import synth
import sa_cases as cases
H = os.path.join(cases.GOLDEN, "hdp")
data = np.array(gzip.open(os.path.join(H, "test_hdp_data.txt.gz"), "rt").read().split(), dtype=np.float64)
dps = np.array(gzip.open(os.path.join(H, "test_hdp_dps.txt.gz"), "rt").read().split(), dtype=np.int64)
s = sa.HdpState.new_tree([-1, 0, 0, 1, 1, 1, 2, 2], 3, (-10.0, 10.0, 50), (0.0, 1.0, 2.0, 10.0), gamma_alpha=[1.0, 1.0, 2.0], gamma_beta=[0.2, 0.2, 0.1])
s.pass_data(data[:8000], dps[:8000])
s.gibbs(3, 30000, 2000, seed=5)
s.write("/tmp/sa_host_asan/tree.hdp")
sa.HdpState("/tmp/sa_host_asan/tree.hdp").write("/tmp/sa_host_asan/tree2.hdp")
assert open("/tmp/sa_host_asan/tree.hdp").read() == open("/tmp/sa_host_asan/tree2.hdp").read()
aln = "/tmp/sa_host_asan/simple_alignment.tsv"
open(aln, "w").write(gzip.open(os.path.join(H, "simple_alignment.tsv.gz"), "rt").read())
nig = sa.hdp_nig_params_from_table(synth.parse_model_table(cases.MODEL_R73)[3])
for layout, gam in ((sa.HDP_LAYOUT_FLAT, [4.0, 20.0]), (sa.HDP_LAYOUT_MULTISET, [1.0, 1.0, 1.0]), (sa.HDP_LAYOUT_MIDDLE_NTS, [1.0, 1.0, 1.0])):
    s = sa.HdpState.new(layout, "ACGT", 6, (0.0, 100.0, 40), nig, gamma=gam)
    s.pass_assignment_file(aln)
    s.gibbs(5, 4000, 500, seed=2)
    s.write("/tmp/sa_host_asan/n.hdp")
    sa.HdpState("/tmp/sa_host_asan/n.hdp")
print("gibbs sweeps under the sanitizers: clean")
PY
