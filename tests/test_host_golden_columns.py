"""The reference's OWN output pins the parameter estimation and the TSV writer's arithmetic.

tests/test_alignments/zymo_C_test_alignments_sm3/tempFiles_alignment/7f22f937-...sm.forward.tsv (600 of its rows are committed
as tests/golden/format/zymo_C_sm3_7f22f937.forward.t300_c300.tsv) is what the reference's signalMachine wrote for the 2-D read
whose .npRead it also ships: tests/test_npReads/ZymoC_ch_1_file1.npRead (event 18 of that file has the noise 0.469887 and the
duration 0.006640 of the golden file's first row; the mean differs by the drift correction).  Seven of a row's sixteen
columns depend only on the read, the model table and the ESTIMATED read parameters -- not on the HMM's transitions or the
band: the drift-corrected event mean, its noise and duration, the scaled model mean E_mean * scale + shift, the scaled model
noise, the descaled event mean and the model mean (writePosteriorProbsFull, impl/signalMachine.c:89-159).  Estimating the
parameters of both strands from the bundled .npRead (signalUtils_estimateNanoporeParams, impl/signalMachineUtils.c:186-225;
impl/nanopore.c:535-954) and applying the writer's formulas must reproduce those columns of all 600 rows to the last printed
digit -- with the library's estimator (host code of the product, no GPU) and with the oracle's.

The same holds for the bundled R9.4 1-D read: tests/test_alignments/ecoli1D_test_alignments_sm3/6deaf971-...sm.forward.tsv is the
reference's output for tests/test_npReads/r9p4_oneD.npRead with models/testModelR9p4_5mer_acegt_template.model (its model-mean
column names that table); its first and last 300 rows are committed next to the Zymo excerpt.

(The posterior column of that file is NOT reproduced by the bundled R7.3 models: the same cells come out with probabilities
that differ by 0.03 on average -- the file was written with model parameters the tree does not hold.)"""
import os

import numpy as np

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

GOLD = os.path.join(cases.GOLDEN, "format", "zymo_C_sm3_7f22f937.forward.t300_c300.tsv")
NPREAD = os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead")
MODEL_C = os.path.join(cases.GOLDEN, "models", "testModelR73_acegot_complement.model")


def _columns(estimate, oracle):
    r = oracle.parse_npread(NPREAD)
    rows = [l.rstrip("\n").split("\t") for l in open(GOLD)]
    out = {}
    for strand, model, evk, mapk, readk in (("t", cases.MODEL_R73, "template_events", "template_strand_event_map", "template_read"),
                                            ("c", MODEL_C, "complement_events", "complement_strand_event_map", "complement_read")):
        alpha, k, t10, tab = synth.parse_model_table(model)
        al = "".join(sorted(alpha))
        ev = r[evk].copy()
        pr = estimate(model, r[mapk], ev, r[readk])
        n = bad = 0
        for g in rows:
            if g[4] != strand:
                continue
            y, kid = int(g[5]), 0
            for ch in g[15]:
                kid = kid * len(al) + al.index(ch)
            e_mean = tab[5 * kid]
            # the estimation has already rescaled the table's noise column once (emissions_signal_scaleNoise); the writer
            # multiplies by scale_sd again (:137-139)
            e_noise = tab[5 * kid + 2] * pr["scale_sd"]
            exp = ["%f" % ev[y, 0], "%f" % ev[y, 1], "%f" % ev[y, 2], "%f" % (e_mean * pr["scale"] + pr["shift"]),
                   "%f" % (e_noise * pr["scale_sd"]),
                   "%f" % ((ev[y, 0] + pr["var"] * e_mean - pr["scale"] * e_mean - pr["shift"]) / pr["var"]), "%f" % e_mean]
            n += 1
            bad += exp != [g[6], g[7], g[8], g[10], g[11], g[13], g[14]]
        out[strand] = (n, bad, pr)
    return out


def test_library_estimator_reproduces_the_reference_output_columns(oracle):
    def estimate(model, emap, ev, read):
        pm = sa.Model.load(model)
        return sa.estimate_params(pm, np.array(pm.table5()).copy(), emap, ev, read)
    res = _columns(estimate, oracle)
    assert res["t"][0] == 300 and res["c"][0] == 300
    assert res["t"][1] == 0 and res["c"][1] == 0, res
    assert abs(res["t"][2]["drift"]) > 1e-3 and abs(res["c"][2]["drift"]) > 1e-3     # the drift correction was exercised


def test_oracle_estimator_reproduces_the_reference_output_columns(oracle):
    def estimate(model, emap, ev, read):
        return oracle.estimate_params(oracle.Model.from_file(model), emap, ev, read)
    res = _columns(estimate, oracle)
    assert res["t"][1] == 0 and res["c"][1] == 0, res


GOLD_1D = os.path.join(cases.GOLDEN, "format", "ecoli1D_sm3_6deaf971.forward.head300_tail300.tsv")
NPREAD_1D = os.path.join(cases.GOLDEN, "npReads", "r9p4_oneD.npRead")
MODEL_1D = os.path.join(cases.GOLDEN, "models", "testModelR9p4_5mer_acegt_template.model")


def test_r9p4_one_d_read_reproduces_the_reference_output_columns(oracle):
    r = oracle.parse_npread(NPREAD_1D)
    rows = [l.rstrip("\n").split("\t") for l in open(GOLD_1D)]
    alpha, k, t10, tab = synth.parse_model_table(MODEL_1D)
    al = "".join(sorted(alpha))
    for which in ("library", "oracle"):
        ev = r["template_events"].copy()
        if which == "library":
            pm = sa.Model.load(MODEL_1D)
            pr = sa.estimate_params(pm, np.array(pm.table5()).copy(), r["template_strand_event_map"], ev, r["template_read"])
        else:
            pr = oracle.estimate_params(oracle.Model.from_file(MODEL_1D), r["template_strand_event_map"], ev, r["template_read"])
        assert abs(pr["drift"]) > 1e-3
        for g in rows:
            assert g[4] == "t"
            y, kid = int(g[5]), 0
            for ch in g[15]:
                kid = kid * len(al) + al.index(ch)
            e_mean, e_noise = tab[5 * kid], tab[5 * kid + 2] * pr["scale_sd"]
            exp = ["%f" % ev[y, 0], "%f" % ev[y, 1], "%f" % ev[y, 2], "%f" % (e_mean * pr["scale"] + pr["shift"]),
                   "%f" % (e_noise * pr["scale_sd"]),
                   "%f" % ((ev[y, 0] + pr["var"] * e_mean - pr["scale"] * e_mean - pr["shift"]) / pr["var"]), "%f" % e_mean]
            assert exp == [g[6], g[7], g[8], g[10], g[11], g[13], g[14]], (which, g, exp)
