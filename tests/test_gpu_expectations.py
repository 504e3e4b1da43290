"""Parity of sa_expect_batch (getExpectationsUsingAnchors, impl/pairwiseAligner.c:2164-2184) against the oracle.

Transition expectations are sums of ~1e5 posteriors accumulated in a different order on the GPU (per checkpoint
group, per lane) than in the reference (cell by cell), so they are compared with a relative tolerance of 1e-9;
the likelihood is a sum of exactly folded totals and must agree to 1e-12 relative; HDP assignments are index
lists and must be identical, in the reference's order.
"""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def _models(oracle, path, nhdp=None):
    alpha, k, t10, tab = synth.parse_model_table(path)
    pm = sa.Model.load(path, nhdp)
    om = oracle.Model(alpha, k, t10, tab)
    if nhdp:
        om.load_hdp(nhdp)
    return pm, om


def _oracle_expect(oracle, om, job, op, ambig=None):
    om.set_read_params(job["scale"], job["shift"], job["var"])
    return oracle.expectations(om, job["ref"], job["events"], job["ax"], job["ay"], op, ambig=ambig)


def test_transition_expectations_gaussian(oracle):
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 700) + cases.synthetic_jobs(cases.MODEL_6MER, 1, 2600, 100)
    trans, lik, assigns = sa.expect_batch(pm, p, jobs)
    for j, job in enumerate(jobs):
        t, l, pos, evs, st = _oracle_expect(oracle, om, job, op)
        assert t[0] > 100 and t[7] == 0.0  # match->match dominates; gapY->gapX is a dead transition
        np.testing.assert_allclose(trans[j], t, rtol=1e-9, atol=1e-12)
        assert abs(lik[j] - l) <= 1e-12 * abs(l)
        assert len(assigns[j]) == 0 and len(pos) == 0  # assignments are an HDP-only feature


def test_expectations_add_to_pseudocounts(oracle):
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 1, 400, 7)
    a, _, _ = sa.expect_batch(pm, p, jobs)
    b, _, _ = sa.expect_batch(pm, p, jobs, pseudocount=0.001)
    np.testing.assert_allclose(b - a, 0.001, rtol=0, atol=1e-9)


def test_expectations_ambiguous_cpg(oracle):
    pm, om = _models(oracle, cases.MODEL_CPG)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, 2, 600, 20, cpg_ambiguous=True)
    trans, lik, _ = sa.expect_batch(pm, p, jobs, ambig=sa.default_ambig({"X": "CE"}))
    for j, job in enumerate(jobs):
        t, l, _, _, _ = _oracle_expect(oracle, om, job, op, ambig=oracle.ambig_map({"X": "CE"}))
        np.testing.assert_allclose(trans[j], t, rtol=1e-9, atol=1e-12)
        assert abs(lik[j] - l) <= 1e-12 * abs(l)


def test_hdp_assignments(oracle):
    pm, om = _models(oracle, cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.1)
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_R73, 2, 600, 40)
    trans, lik, assigns = sa.expect_batch(pm, p, jobs)
    for j, job in enumerate(jobs):
        t, l, pos, evs, st = _oracle_expect(oracle, om, job, op)
        assert len(pos) > 20  # the fixture HDP only has densities for the k-mers it observed (read ends here)
        # device log() in the HDP emission: same tolerance class as the posterior test
        np.testing.assert_allclose(trans[j], t, rtol=1e-6, atol=1e-9)
        assert abs(lik[j] - l) <= 1e-9 * abs(l)
        ev = np.asarray(job["events"], dtype=np.float64)
        mean = ev if ev.ndim == 1 else ev[:, 0]
        got = list(zip(assigns[j][:, 0].tolist(), mean[assigns[j][:, 1]].tolist()))
        exp = list(zip(pos.tolist(), evs.tolist()))
        if got != exp:
            # only assignments whose probability sits on the threshold may differ
            assert abs(len(got) - len(exp)) <= 2 and len(set(got) ^ set(exp)) <= 2
