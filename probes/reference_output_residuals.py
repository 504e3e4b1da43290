"""Where do the CPU restatement's posteriors differ from the ones the REFERENCE printed?  

For the two reference output files pinned by tests/test_oracle_reference_outputs.py: every reference row (x, y, p) is located in
THIS run's band (built from this run's anchors -- lastz's for the Zymo read, an alignment rebuilt from the rows for the E. coli
read -- where the reference's came from bwa), its distance to the band's edge on its anti-diagonal and to the nearest anchor is
taken, and |dp| is tabulated against that distance.  CPU only.  usage: python probes/reference_output_residuals.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sa_oracle_py as oracle   # noqa: E402
from signalalign_amd import synth           # noqa: E402
import sa_cases as cases                     # noqa: E402

EXP = os.path.join(cases.GOLDEN, "expected")
EXPANSION = 50


def analyse(name, gold, mine, ax, ay, lX, lY, x_off, y_off):
    """gold / mine: {(x, y): p} in file coordinates; anchors and band in region coordinates (x - x_off, y - y_off)."""
    xmyL, xmyR = oracle.band(ax, ay, lX, lY, EXPANSION)
    anchors = np.stack([np.asarray(ax) + 1, np.asarray(ay) + 1], axis=1) if len(ax) else np.zeros((0, 2))   # matrix coordinates
    rows = []
    for (x, y), p in gold.items():
        mx, my = x - x_off + 1, y - y_off + 1            # matrix coordinates of the cell
        d, xmy = mx + my, mx - my
        inside = 0 <= d < len(xmyL) and xmyL[d] <= xmy <= xmyR[d]
        edge = min(xmy - xmyL[d], xmyR[d] - xmy) // 2 if inside else -1
        a_dist = int(np.min(np.maximum(np.abs(anchors[:, 0] - mx), np.abs(anchors[:, 1] - my)))) if len(anchors) else -1
        dp = abs(mine[(x, y)] - p) if (x, y) in mine else None
        rows.append((edge, a_dist, inside, dp, p))
    n = len(rows)
    found = [r for r in rows if r[3] is not None]
    missing = [r for r in rows if r[3] is None]
    out = {"read": name, "reference_rows": n, "found": len(found), "missing": len(missing),
           "missing_outside_this_band": sum(1 for r in missing if not r[2]),
           "missing_inside_band_max_p": max([r[4] for r in missing if r[2]], default=None),
           "rows_beyond_1e-4": sum(1 for r in found if r[3] > 1e-4)}
    bins = [(0, 2), (3, 5), (6, 10), (11, 20), (21, 60)]
    tab = []
    for lo, hi in bins:
        sel = [r for r in found if lo <= r[0] <= hi]
        if sel:
            dps = np.array([r[3] for r in sel])
            tab.append({"cells_from_band_edge": "%d-%d" % (lo, hi), "rows": len(sel), "within_1e-4": float((dps <= 1e-4).mean()),
                        "within_1e-5": float((dps <= 1e-5).mean()), "median": float(np.median(dps)), "max": float(dps.max())})
    out["by_distance_to_band_edge"] = tab
    tab2 = []
    for lo, hi in [(0, 5), (6, 15), (16, 40), (41, 10 ** 6)]:
        sel = [r for r in found if lo <= r[1] <= hi]
        if sel:
            dps = np.array([r[3] for r in sel])
            tab2.append({"cells_from_nearest_anchor": "%d-%s" % (lo, hi if hi < 10 ** 6 else "inf"), "rows": len(sel),
                         "within_1e-4": float((dps <= 1e-4).mean()), "median": float(np.median(dps)), "max": float(dps.max())})
    out["by_distance_to_nearest_anchor"] = tab2
    # consecutive runs of rows beyond 1e-4 along the read: are the differences local stretches?
    bad = sorted((x + y) for (x, y), p in gold.items() if (x, y) in mine and abs(mine[(x, y)] - p) > 1e-4)
    runs, start, prev = [], None, None
    for d in bad:
        if start is None or d - prev > 40:
            if start is not None:
                runs.append((start, prev))
            start = d
        prev = d
    if start is not None:
        runs.append((start, prev))
    out["stretches_of_rows_beyond_1e-4"] = {"count": len(runs), "diagonal_ranges": runs[:12]}
    return out


def zymo():
    z = np.load(os.path.join(EXP, "reference_output_zymo2d.npz"))
    t = z["strand"] == "t"
    gold = {(int(x), int(y)): float(p) for x, y, p in zip(z["x"][t], z["y"][t], z["p"][t])}
    r = oracle.parse_npread(os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead"))
    ref = "".join(open(os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta")).read().split("\n")[1:])
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    s2, e2, s1, e1 = int(cig[2]), int(cig[3]), int(cig[6]), int(cig[7])
    ops = [({"M": 0, "D": 1, "I": 2}[cig[i]], int(cig[i + 1])) for i in range(10, len(cig), 2)]
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    om = oracle.Model(alpha, k, t10, tab, emission=oracle.EM_TWODIST_DESCALED)
    ev = r["template_events"].copy()
    pr = oracle.estimate_params(om, r["template_strand_event_map"], ev, r["template_read"])
    gx, gy = oracle.guide_to_anchors(s1, e1, 1, s2, ops, 14)
    em = r["template_event_map"]
    ax, ay = oracle.remap_anchors(gx, gy, em, s2)
    lo, hi = int(em[s2]), int(em[e2 - 1])
    om.set_read_params(pr["scale"], pr["shift"], pr["var"])
    target = ref[s1:e1]
    pairs = oracle.align(om, target, ev[lo:hi], ax, ay, oracle.Params(0.01, EXPANSION, 100, 1000, 3000 * 3000, 14))
    mine = {(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in pairs}
    return analyse("ZymoC_ch_1_file1 (2-D read, template strand; anchors from the reference's lastz cigar)", gold, mine, ax, ay,
                   len(target) - (k - 1), hi - lo, s1, lo)


def ecoli():
    gold, window, r, (s1, e1, s2, e2), ops = cases.reference_output_ecoli1d_inputs(oracle)
    read, em = r["template_read"], r["template_strand_event_map"]
    alpha, k, t10, tab = synth.parse_model_table(os.path.join(cases.GOLDEN, "models", "testModelR9p4_5mer_acegt_template.model"))
    om = oracle.Model(alpha, k, t10, tab, emission=oracle.EM_TWODIST_DESCALED)
    ev = r["template_events"].copy()
    pr = oracle.estimate_params(om, em, ev, read)
    gx, gy = oracle.guide_to_anchors(s1, e1, 1, s2, ops, 14)
    ax, ay = oracle.remap_anchors(gx, gy, em, s2)
    lo, hi = int(em[s2]), int(em[e2 - 1])
    om.set_read_params(pr["scale"], pr["shift"], pr["var"])
    target = window[s1:e1]
    pairs = oracle.align(om, target, ev[lo:hi], ax, ay, oracle.Params(0.01, EXPANSION, 100, 1000, 3000 * 3000, 14))
    mine = {(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in pairs}
    return analyse("r9p4_oneD (1-D read; guide alignment rebuilt from the reference's own rows)", gold, mine, ax, ay,
                   len(target) - (k - 1), hi - lo, s1, lo)


if __name__ == "__main__":
    print(json.dumps([zymo(), ecoli()], indent=1))
