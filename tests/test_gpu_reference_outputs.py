"""The HIP path against posteriors the reference itself wrote: the Zymo 2-D read's template strand (fixtures and background:
tests/test_oracle_reference_outputs.py).  The file was written by a build whose state machine carried the two-distribution
emission, which the reference-ordered memory-resident kernels (SA_FLAG_EXACT) and, since round 6, the register kernels (the
default for a read with one path per cell: k_fwd_fast_two / k_bwd_fast_two) offer behind sa_model_set_emission: the
reference-ordered kernels' pairs are bit-identical to the CPU restatement's, the register kernels' within 1e-5 of them, and all
sit on the reference's printed posteriors (median |dp| <= 2e-6, nine rows in ten within 1e-4, the rest being the guide alignment
bwa made and lastz did not)."""
import json
import os

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def test_gpu_reproduces_the_reference_posteriors_of_the_zymo_read(oracle):
    z = np.load(os.path.join(cases.GOLDEN, "expected", "reference_output_zymo2d.npz"))
    t = z["strand"] == "t"
    gold = {(int(x), int(y)): float(p) for x, y, p in zip(z["x"][t], z["y"][t], z["p"][t])}
    r = oracle.parse_npread(os.path.join(cases.GOLDEN, "npReads", "ZymoC_ch_1_file1.npRead"))
    ref = "".join(open(os.path.join(cases.GOLDEN, "sequences", "zymo_sequence.fasta")).read().split("\n")[1:])
    cig = json.load(open(os.path.join(cases.GOLDEN, "cigars", "zymoC_lastz_anchors.json")))["calls"][0]["cigars"][0].split()
    s2, e2, s1, e1 = int(cig[2]), int(cig[3]), int(cig[6]), int(cig[7])
    ops = [({"M": 0, "D": 1, "I": 2}[cig[i]], int(cig[i + 1])) for i in range(10, len(cig), 2)]
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    # the product's own host side: parameter estimation (rescales the noise columns of its table copy), anchors
    pm0 = sa.Model.load(cases.MODEL_R73)
    ev = r["template_events"].copy()
    t5 = np.array(pm0.table5()).copy()
    pr = sa.estimate_params(pm0, t5, r["template_strand_event_map"], ev, r["template_read"])
    gx, gy = sa.guide_to_anchors(s1, e1, 1, s2, ops, 14)
    em = r["template_event_map"]
    ax, ay = sa.remap_anchors(gx, gy, em, s2)
    lo, hi = int(em[s2]), int(em[e2 - 1])
    pm = sa.Model.create(alpha, k, t10, t5)
    pm.set_emission(1)                                   # SA_EMISSION_TWO_DIST
    job = dict(ref=ref[s1:e1], events=np.ascontiguousarray(ev[lo:hi]), ax=ax, ay=ay, scale=pr["scale"], shift=pr["shift"], var=pr["var"])
    p = sa.default_params(threshold=0.01, expansion=50, trace_back=100)
    b = sa.Batch(pm, p, [job], flags=sa.FLAG_EXACT)
    b.run()
    got = b.pairs(0)
    assert b.stats().n_fast_regions == 0                 # the reference-ordered kernels
    b.close()
    bf = sa.Batch(pm, p, [job])                          # ... and the register kernels (round 6), checked below
    bf.run()
    got_fast = bf.pairs(0)
    assert bf.stats().n_fast_regions == bf.stats().n_regions >= 1
    bf.close()
    # the CPU restatement with the same emission: bit-identical
    om = oracle.Model(alpha, k, t10, tab, emission=oracle.EM_TWODIST_DESCALED)
    ev_o = r["template_events"].copy()
    pr_o = oracle.estimate_params(om, r["template_strand_event_map"], ev_o, r["template_read"])
    om.set_read_params(pr_o["scale"], pr_o["shift"], pr_o["var"])
    exp = oracle.align(om, ref[s1:e1], ev_o[lo:hi], ax, ay, cases.oracle_params(oracle, p))
    assert len(got) == len(exp)
    for f in ("x", "y", "kmer_id", "prob_e7"):
        assert np.array_equal(got[f], exp[f]), f
    # ... and on the reference's printed posteriors
    mine = {(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in got}
    common = set(mine) & set(gold)
    d = np.array([abs(mine[k_] - gold[k_]) for k_ in common])
    assert len(common) >= 0.97 * len(gold) and np.median(d) <= 2e-6 and (d <= 1e-4).mean() >= 0.9
    found, _, within_rel, _, beyond = cases.reference_residual(mine, gold)   # (the residual is one factor per checkpoint group)
    assert within_rel >= 0.99 and all(row[0] <= 60 for row in beyond), (within_rel, beyond[:5])
    # the register kernels: within 1e-5 of the reference-ordered ones, and on the reference's printed posteriors as well
    w, lonely = cases.compare_pairs(got_fast, got, 100, p.threshold)
    assert w <= 10 and lonely <= 2 and cases.same_order(got_fast, got)
    mine_f = {(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in got_fast}
    common_f = set(mine_f) & set(gold)
    d_f = np.array([abs(mine_f[k_] - gold[k_]) for k_ in common_f])
    assert len(common_f) >= 0.97 * len(gold) and np.median(d_f) <= 2e-6 and (d_f <= 1e-4).mean() >= 0.9
    # a dense event vector cannot carry the noise
    with pytest.raises(sa.SaError):
        sa.Batch(pm, p, [dict(job, events=np.ascontiguousarray(job["events"][:, 0]))])


def test_gpu_reproduces_the_reference_posteriors_of_the_r9p4_read(oracle):
    """The same for the bundled R9.4 1-D read (10.9k events, 5-mer ACEGT model; window and guide alignment rebuilt from the
    reference's rows: sa_cases.reference_output_ecoli1d_inputs): bit-identical to the restatement, on the reference's printed
    posteriors within what the cruder guide alignment allows."""
    gold, window, r, (s1, e1, s2, e2), ops = cases.reference_output_ecoli1d_inputs(oracle)
    model = os.path.join(cases.GOLDEN, "models", "testModelR9p4_5mer_acegt_template.model")
    alpha, k, t10, tab = synth.parse_model_table(model)
    em, read = r["template_strand_event_map"], r["template_read"]
    pm0 = sa.Model.load(model)
    ev = r["template_events"].copy()
    t5 = np.array(pm0.table5()).copy()
    pr = sa.estimate_params(pm0, t5, em, ev, read)
    gx, gy = sa.guide_to_anchors(s1, e1, 1, s2, ops, 14)
    ax, ay = sa.remap_anchors(gx, gy, em, s2)
    lo, hi = int(em[s2]), int(em[e2 - 1])
    pm = sa.Model.create(alpha, k, t10, t5)
    pm.set_emission(1)
    p = sa.default_params(threshold=0.01, expansion=50, trace_back=100)
    job = dict(ref=window[s1:e1], events=np.ascontiguousarray(ev[lo:hi]), ax=ax, ay=ay, scale=pr["scale"], shift=pr["shift"], var=pr["var"])
    b = sa.Batch(pm, p, [job], flags=sa.FLAG_EXACT)
    b.run()
    got = b.pairs(0)
    b.close()
    bf = sa.Batch(pm, p, [job])                          # the register kernels (round 6)
    bf.run()
    got_fast = bf.pairs(0)
    assert bf.stats().n_fast_regions == bf.stats().n_regions >= 1
    bf.close()
    w, lonely = cases.compare_pairs(got_fast, got, 100, p.threshold)
    assert w <= 10 and lonely <= 4 and cases.same_order(got_fast, got)
    om = oracle.Model(alpha, k, t10, tab, emission=oracle.EM_TWODIST_DESCALED)
    ev_o = r["template_events"].copy()
    pr_o = oracle.estimate_params(om, em, ev_o, read)
    om.set_read_params(pr_o["scale"], pr_o["shift"], pr_o["var"])
    exp = oracle.align(om, window[s1:e1], ev_o[lo:hi], ax, ay, cases.oracle_params(oracle, p))
    assert len(got) == len(exp) > 10000
    for f in ("x", "y", "kmer_id", "prob_e7"):
        assert np.array_equal(got[f], exp[f]), f
    mine = {(int(q["x"]) + s1, int(q["y"]) + lo): int(q["prob_e7"]) / 1e7 for q in got}
    common = set(mine) & set(gold)
    d = np.array([abs(mine[k_] - gold[k_]) for k_ in common])
    assert len(common) >= 0.8 * len(gold) and np.median(d) <= 5e-6 and (d <= 1e-4).mean() >= 0.8
    assert cases.reference_residual(mine, gold)[2] >= 0.94
