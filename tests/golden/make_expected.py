#!/usr/bin/env python3
"""Regenerates tests/golden/expected/*.npz: what the CPU oracle (oracle/sa_oracle.c) computes for the bundled reads.

Inputs: the reference's own fixture files under tests/golden (npReads, models).  Each read is aligned to its own
template read (one M run, trim 14) -- the substitution for the missing E. coli references that SURVEY section 8(c)
prescribes -- with the signalMachine defaults (-x 50 -D 0.01 -g 100).  Stored per case: the estimated read parameters,
the anchors, and the aligned pairs (x, y, path, kmer_id, prob_e7) in the reference's output order.

These files pin BOTH sides: tests/test_oracle_kats.py checks that the oracle still reproduces them bit for bit, and
tests/test_gpu_parity.py checks the HIP path against them without running the oracle.
Usage: python tests/golden/make_expected.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sa_cases as cases  # noqa: E402
from oracle import sa_oracle_py as oracle  # noqa: E402

CASES = [
    ("r9_5mer", "c2925_ecoli_ch34_read1023.npRead", cases.MODEL_5MER, None),
    ("r94_6mer", "r9p4_oneD.npRead", cases.MODEL_6MER, None),
    ("r73_acegot", "ZymoC_ch_1_file1.npRead", cases.MODEL_R73, None),
    ("r73_acegot_hdp", "ZymoC_ch_1_file1.npRead", cases.MODEL_R73, cases.NHDP),
]


def compute(name, npread, model_path, nhdp):
    job = cases.npread_job(oracle, npread, model_path)
    om = oracle.Model.from_file(model_path)
    threshold = 0.01
    if nhdp:
        om.load_hdp(nhdp)
        om.set_to_hdp_expected_values()
        threshold = 0.05
    om.set_read_params(job["scale"], job["shift"], job["var"])
    p = oracle.default_params(threshold=threshold)
    pairs = oracle.align(om, job["ref"], job["events"], job["ax"], job["ay"], p)
    return dict(threshold=np.float64(threshold), scale=np.float64(job["scale"]), shift=np.float64(job["shift"]),
                var=np.float64(job["var"]), ax=np.asarray(job["ax"], dtype=np.int64), ay=np.asarray(job["ay"], dtype=np.int64),
                x=pairs["x"].astype(np.int32), y=pairs["y"].astype(np.int32), path=pairs["path"].astype(np.int32),
                kmer_id=pairs["kmer_id"].astype(np.int32), prob_e7=pairs["prob_e7"].astype(np.int64))


# The reference's whole-read test inputs (tests/stateMachineTests.c:842-983): ZymoC x ZymoRef with the anchors of the
# reference's lastz subprocess (tests/zymo_wholeread.py), banding defaults of pairwiseAlignmentBandingParameters_construct,
# read parameters from the .npRead header.  "zymo_lastz_hdp" IS the reference's test_sm3Hdp_getAlignedPairsWithBanding job
# (its emission is the one signalMachine --sm3Hdp uses): 1217 pairs.  The Gaussian variants run the same jobs with the
# emission signalMachine installs (MeanOnly, impl/signalMachine.c:347-349) instead of the unit tests' two-distribution
# one, so their counts are the oracle's, not the reference's 1076 / 7349 (those are asserted in test_oracle_kats.py).
ZYMO_CASES = ["zymo_lastz_C", "zymo_lastz_E", "zymo_lastz_O", "zymo_lastz_L", "zymo_lastz_hdp"]


def zymo_job(name):
    import zymo_wholeread as z
    r = z.read_fixture()
    ax, ay = z.remapped_anchors()
    tp = r["template_params"]
    hdp = name.endswith("_hdp")
    ref = r["ref"] if hdp else r["ref"].replace("C", name[-1])
    ev = z.hdp_test_events(r) if hdp else r["template_events"]
    b = z.BANDING
    return dict(ref=ref, events=np.ascontiguousarray(ev), ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"],
                var=tp["var"]), dict(b, threshold=0.1 if hdp else b["threshold"]), hdp


def compute_zymo(name):
    job, b, hdp = zymo_job(name)
    om = oracle.Model.from_file(cases.MODEL_R73)
    if hdp:
        om.load_hdp(cases.NHDP)       # getHdpStateMachine: no setModelToHdpExpectedValues (impl/stateMachine.c:1778)
    om.set_read_params(job["scale"], job["shift"], job["var"])
    p = oracle.Params(b["threshold"], b["expansion"], b["trace_back"], b["min_diags"], b["split"], 14)
    pairs = oracle.align(om, job["ref"], job["events"], job["ax"], job["ay"], p)
    return dict(threshold=np.float64(b["threshold"]), x=pairs["x"].astype(np.int32), y=pairs["y"].astype(np.int32),
                path=pairs["path"].astype(np.int32), kmer_id=pairs["kmer_id"].astype(np.int32),
                prob_e7=pairs["prob_e7"].astype(np.int64))


if __name__ == "__main__":
    out = os.path.join(HERE, "expected")
    os.makedirs(out, exist_ok=True)
    for name, npread, model_path, nhdp in CASES:
        d = compute(name, npread, model_path, nhdp)
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, len(d["x"]), "pairs")
    for name in ZYMO_CASES:
        d = compute_zymo(name)
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, len(d["x"]), "pairs")
