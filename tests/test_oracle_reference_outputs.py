"""The CPU restatement against posteriors the REFERENCE ITSELF wrote.

Two of the output files the reference ships turned out to be signalMachine's results for reads whose .npRead it also ships
(tests/golden/make_reference_output_fixtures.py names them; tests/test_host_golden_columns.py pins the parameter columns).  Their
posterior column is reproduced by the restatement when it uses the two-distribution emission -- Gaussian on the descaled event
mean times inverse Gaussian on the event noise, emissions_signal_strawManGetKmerEventMatchProbWithDescaling,
impl/stateMachine.c:607-650 -- i.e. the files were written by a build whose state machine carried that emission (today's
signalMachine installs the MeanOnly variant, :557-605; with it the same cells come out 0.03 apart on average).  Everything else is
the path under test: parameter estimation and drift correction, anchors from the guide alignment, the band, forward and backward
sweeps with the reference's logAdd, periodic traceback, total probabilities and posteriors.

What cannot be the same is the guide alignment (the files were made with bwa's; here the Zymo read uses the reference's lastz
cigar and the E. coli read an alignment rebuilt from the rows themselves), so cells near band edges and uncertain stretches differ.
The bars: at least 97 % of the reference's rows are found, half of them agree to the printed precision (median |dp| <= 2e-6) and
nine in ten to 1e-4.

Round 4 (probes/reference_output_residuals.py, sa_cases.reference_residual): the rows beyond 1e-4 are NOT at band edges -- they
share one multiplicative factor per group of diagonals (|log| <= 1.5e-3), the total probability the reference refreshes every tenth
diagonal of a traceback, whose phase follows the guide alignment.  Relative to that factor 99.5 % of the Zymo rows agree."""
import json
import os

import numpy as np

from signalalign_amd import synth

import sa_cases as cases

EXP = os.path.join(cases.GOLDEN, "expected")


def _compare(mine, gold):
    found, median, within_rel, within_abs, beyond = cases.reference_residual(mine, gold)
    return found, median, within_abs, within_rel, beyond


def test_zymo_two_d_template_posteriors_of_the_reference(oracle):
    z = np.load(os.path.join(EXP, "reference_output_zymo2d.npz"))
    t = z["strand"] == "t"
    gold = {(int(x), int(y)): float(p) for x, y, p in zip(z["x"][t], z["y"][t], z["p"][t])}
    r = oracle.parse_npread(os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead"))
    ref = "".join(open(os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta")).read().split("\n")[1:])
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    s2, e2, s1, e1 = int(cig[2]), int(cig[3]), int(cig[6]), int(cig[7])
    ops = [({"M": 0, "D": 1, "I": 2}[cig[i]], int(cig[i + 1])) for i in range(10, len(cig), 2)]
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    res = {}
    for name, emission in (("two_dist", oracle.EM_TWODIST_DESCALED), ("mean_only", oracle.EM_MEANONLY_DESCALED)):
        om = oracle.Model(alpha, k, t10, tab, emission=emission)
        ev = r["template_events"].copy()
        pr = oracle.estimate_params(om, r["template_strand_event_map"], ev, r["template_read"])
        gx, gy = oracle.guide_to_anchors(s1, e1, 1, s2, ops, 14)
        em = r["template_event_map"]
        ax, ay = oracle.remap_anchors(gx, gy, em, s2)
        lo, hi = int(em[s2]), int(em[e2 - 1])
        om.set_read_params(pr["scale"], pr["shift"], pr["var"])
        pairs = oracle.align(om, ref[s1:e1], ev[lo:hi], ax, ay, oracle.Params(0.01, 50, 100, 1000, 3000 * 3000, 14))
        res[name] = _compare({(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in pairs}, gold)
    found, median, within, within_rel, beyond = res["two_dist"]
    assert found >= 0.97 and median <= 2e-6 and within >= 0.9, res
    # what the residual IS (sa_cases.reference_residual): one factor per group of diagonals, from the total the reference refreshes
    # every tenth diagonal -- measured 99.5 % of the rows within 1e-3 p; the three rows beyond 1.5e-3 p sit on diagonals 40-44,
    # where the lastz guide alignment starts differently from bwa's
    assert within_rel >= 0.99 and all(row[0] <= 60 for row in beyond), (within_rel, beyond[:5])
    assert res["mean_only"][1] > 1e-3            # the emission signalMachine installs today does not reproduce the file
    # the cause, shown on the same read: the reference-ordered total of the UN-BANDED matrix moves from diagonal to diagonal by as
    # much as the rows' factors do
    om = oracle.Model(alpha, k, t10, tab, emission=oracle.EM_TWODIST_DESCALED)
    ev = r["template_events"].copy()
    pr = oracle.estimate_params(om, r["template_strand_event_map"], ev, r["template_read"])
    om.set_read_params(pr["scale"], pr["shift"], pr["var"])
    em = r["template_event_map"]
    _, _, diag, _ = oracle.kat_unbanded(om, ref[s1:e1], ev[int(em[s2]):int(em[e2 - 1])], 0.01)
    d = diag[60:-60]
    step10 = np.abs(d[10:] - d[:-10])
    assert 2e-4 < np.percentile(step10, 95) < 2e-3 and 5e-4 < step10.max() < 5e-3, (np.percentile(step10, 95), step10.max())


def test_r9p4_one_d_posteriors_of_the_reference(oracle):
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
    pairs = oracle.align(om, window[s1:e1], ev[lo:hi], ax, ay, oracle.Params(0.01, 50, 100, 1000, 3000 * 3000, 14))
    found, median, within, within_rel, beyond = _compare({(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in pairs}, gold)
    # (the rebuilt guide alignment is cruder than bwa's: fewer rows are found and fewer agree than for the Zymo read.  Measured:
    # 95.3 % of the found rows within the relative bar of sa_cases.reference_residual; the rest lie in stretches where the rebuilt
    # alignment takes another path)
    assert found >= 0.8 and median <= 5e-6 and within >= 0.8 and within_rel >= 0.94, (found, median, within, within_rel)
