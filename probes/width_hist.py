"""Band geometry of the `realistic` bench workload (host only): diagonal widths, how the first column moves, and the lane
utilisation a K-columns-per-lane register kernel would have (64 K cell slots per diagonal)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import signalalign_amd as sa
from signalalign_amd import synth, _capi

MODEL = os.path.join(ROOT, "tests", "golden", "models", "testModelR9.4_5mer_acgt_template.model")
if not os.path.exists(MODEL):
    import glob
    MODEL = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "models", "*R9.4*6mer*template*.model")))[0]
import bench
MODEL = bench.MODEL
alpha, k, t10, tab = synth.parse_model_table(MODEL)
pm = sa.Model.load(MODEL, None)
params = sa.default_params(threshold=0.01, expansion=50, trace_back=100)
toks = open(os.path.join(ROOT, "tests", "golden", "cigars", "ecoli_minus_strand.cigar")).read().split()[10:]
runs = [(toks[i], int(toks[i + 1])) for i in range(0, len(toks), 2)]
def thin(job, idx):
    keep = np.zeros(len(job["ax"]), dtype=bool)
    pos, r = 0, (7 * int(idx)) % len(runs)
    while pos < len(keep):
        op, ln = runs[r % len(runs)]
        r += 1
        if op == "M":
            if ln > 28:
                keep[pos + 14: min(pos + ln - 14, len(keep))] = True
            pos += ln
        elif op == "D":
            pos += ln
    job["ax"], job["ay"] = job["ax"][keep], job["ay"][keep]
hist = np.zeros(1024, dtype=np.int64)
maxw = []
dxs = np.zeros(16, dtype=np.int64)
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for i in range(nreads):
    job = synth.make_read(i, 5000, alpha, k, tab)
    thin(job, i)
    info, reg, rows, segs = _capi.plan_describe(pm, params, job)
    off = 0
    for r in reg:
        n = int((r[2] - r[0]) + (r[3] - r[1]) + 1)
        rr = rows[off:off + n]; off += n
        w = (rr[:, 2] - rr[:, 1]) // 2 + 1          # rows: region, xmyL, xmyR
        maxw.append(int(w.max()))
        hist += np.bincount(np.minimum(w, 1023), minlength=1024)
        xl = (np.arange(n) + rr[:, 1]) // 2
        dx = np.diff(xl)
        dxs += np.bincount(np.clip(dx + 8, 0, 15), minlength=16)
tot = hist.sum(); cells = (hist * np.arange(1024)).sum()
print("reads", nreads, "regions", len(maxw), "diagonals", tot, "cells/diag %.1f" % (cells / tot))
print("max width per region:", sorted(maxw)[::max(1, len(maxw) // 12)])
print("first-column step histogram (dx -> count):", {int(i - 8): int(c) for i, c in enumerate(dxs) if c})
c = np.cumsum(hist) / tot
for q in (64, 96, 127, 160, 191, 224, 255, 319, 383):
    print("w <= %d: %.3f of diagonals" % (q, c[q]))
for K in (2, 3, 4, 5, 6):
    ok = np.array(maxw) <= 64 * K - 1
    print("K=%d: regions that fit %.2f; utilisation if all diagonals ran at this K: %.2f" % (K, ok.mean(), cells / (64.0 * K * tot)))
