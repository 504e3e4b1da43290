"""(probe -- run with SA_LIBRARY=probes/_variants/lib_ringp.so, see probes/ringp/README.md)
The packed ring sweeps (probes/ringp/sa_ringp.inc, round 6): four regions with several paths per cell share a
workgroup -- a wave per region for the first 64 cell-paths of a diagonal, the tails of all four packed into shared waves.

The arithmetic of a cell-path is the same sequence of logAdds as in the one-region-per-workgroup kernels (sa_ring.inc), so every
result -- rows, order, prob_e7 -- must be IDENTICAL to theirs (SA_RING_PACKED=0), whatever the composition of a group: regions of
different lengths, a last group that is short, rows of at most 64 cell-paths beside rows of 120, a batch with one region.  The
oracle is consulted for a sample of the reads (every read of these shapes is checked against it in tests/test_gpu_parity.py).
Reference semantics: impl/pairwiseAligner.c:723-801 (hdCell_construct2), impl/stateMachine.c:1306-1369."""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import sa_cases as cases  # noqa: E402

pytestmark = pytest.mark.gpu

TOL_E7 = 100


def _run(pm, params, jobs, ambig, flags=0):
    b = sa.Batch(pm, params, jobs, ambig=ambig, flags=flags)
    b.run()
    out = [b.pairs(j) for j in range(len(jobs))]
    st = b.stats()
    b.close()
    return out, st


def _cpg_jobs():
    jobs = []
    # lengths between 300 and 2600 events, 23 reads (five full groups and one of three), dense and sparse ambiguity letters
    for i, n_ev in enumerate([2600, 300, 1900, 1200, 700, 2400, 2300, 950, 1500, 1450, 400, 2100, 1800, 640, 1000, 2200, 330, 1250, 1700,
                              860, 2000, 520, 1600]):
        kw = {"cpg_ambiguous": True}
        if i % 5 == 3:
            kw["cpg_every"] = 7
        jobs += cases.synthetic_jobs(cases.MODEL_CPG, 1, n_ev, 9100 + 17 * i, **kw)
    # two reads whose ends are not ragged (sa_job_t.ends), one with anchors thinned so that the band widens (rows beyond 128
    # cell-paths: that region stays on the unpacked kernels, in the same batch)
    jobs[4] = dict(jobs[4], ragged=(0, 0))
    jobs[9] = dict(jobs[9], ragged=(0, 1))
    wide = dict(jobs[12])
    keep = np.zeros(len(wide["ax"]), dtype=bool)
    keep[::29] = True
    wide["ax"], wide["ay"] = wide["ax"][keep], wide["ay"][keep]
    jobs.append(wide)
    return jobs


@pytest.mark.parametrize("threshold", [0.01, 0.2])
def test_packed_forward_sweep_gives_the_bytes_of_the_unpacked_one(oracle, monkeypatch, threshold):
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_CPG)
    pm = sa.Model.load(cases.MODEL_CPG)
    om = oracle.Model(alpha, k, t10, tab)
    p = sa.default_params(threshold=threshold)
    op = cases.oracle_params(oracle, p)
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    jobs = _cpg_jobs()
    got, st = _run(pm, p, jobs, amb_p)
    assert st.n_ring_regions == len(jobs)
    monkeypatch.setenv("SA_RING_PACKED", "0")
    ref, _ = _run(pm, p, jobs, amb_p)
    monkeypatch.delenv("SA_RING_PACKED")
    for j in range(len(jobs)):
        assert np.array_equal(got[j], ref[j]), (j, len(got[j]), len(ref[j]))
    # the shared waves' unprepared path (a second chunk of tails with one shared wave) and idle shared waves (four of them)
    for shared in ("1", "4"):
        monkeypatch.setenv("SA_RINGP_SHARED", shared)
        again, _ = _run(pm, p, jobs, amb_p)
        monkeypatch.delenv("SA_RINGP_SHARED")
        for j in range(len(jobs)):
            assert np.array_equal(again[j], ref[j]), (shared, j)
    # batches of one, two and five regions (groups that are never full)
    for sub in ([jobs[0]], jobs[3:5], jobs[5:10]):
        one, _ = _run(pm, p, sub, amb_p)
        monkeypatch.setenv("SA_RING_PACKED", "0")
        two, _ = _run(pm, p, sub, amb_p)
        monkeypatch.delenv("SA_RING_PACKED")
        for j in range(len(sub)):
            assert np.array_equal(one[j], two[j]), (len(sub), j)
    # ... and the oracle on a sample
    worst = 0
    for j in (0, 1, 4, 9, 22):
        job = jobs[j]
        exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2 and cases.same_order(got[j], exp), j
    assert worst <= 10


def test_packed_forward_sweep_hdp_emissions(oracle, monkeypatch):
    """--sm3Hdp with variant positions: the packed sweep reads one emission per cell-path from the region's plane"""
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.05)
    amb_p = sa.default_ambig({"X": "CE"})
    jobs = []
    for i, n_ev in enumerate([1300, 500, 900, 1100, 700, 1000]):
        for job in cases.hdp_jobs(1, n_ev, 400 + 13 * i, table5=pm.table5()):
            job["ref"] = job["ref"][:10] + job["ref"][10:-10].replace("CG", "XG") + job["ref"][-10:]
            jobs.append(job)
    got, st = _run(pm, p, jobs, amb_p)
    assert st.n_ring_regions >= 5
    monkeypatch.setenv("SA_RING_PACKED", "0")
    ref, _ = _run(pm, p, jobs, amb_p)
    monkeypatch.delenv("SA_RING_PACKED")
    for j in range(len(jobs)):
        assert np.array_equal(got[j], ref[j]), (j, len(got[j]), len(ref[j]))
