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
    """Several paths per cell (EM training of a methylation model): round 4 runs these regions on the ring kernels' expectation
    variant (k_bwd_ring<EXPECT>: forward states from the planes, legal predecessors from the per-path records) -- against the
    oracle and against the memory-resident checker; dense and sparse anchors, short tracebacks, a three-way code."""
    pm, om = _models(oracle, cases.MODEL_CPG)
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, 3, 1400, 20, cpg_ambiguous=True)
    sparse = cases.realistic_anchor_jobs(cases.MODEL_CPG, 2, 1200, 77)
    jobs += [dict(j, ref=j["ref"].replace("CG", "XG")) for j in sparse]
    jobs += cases.synthetic_jobs(cases.MODEL_CPG, 1, 60, 5, cpg_ambiguous=True)
    for kw in (dict(), dict(expansion=20, trace_back=30, min_diags=150)):
        p = sa.default_params(**kw)
        op = cases.oracle_params(oracle, p)
        trans, lik, _ = sa.expect_batch(pm, p, jobs, ambig=amb_p)
        st = sa.expect_last_stats()
        assert st.n_ring_regions >= len(jobs) - 1, (st.n_regions, st.n_fast_regions, st.n_ring_regions)
        chk, chk_lik, _ = sa.expect_batch(pm, p, jobs, ambig=amb_p, flags=sa.FLAG_FORCE_GENERIC)
        assert sa.expect_last_stats().n_ring_regions == 0
        for j, job in enumerate(jobs):
            t, l, _, _, _ = _oracle_expect(oracle, om, job, op, ambig=amb_o)
            np.testing.assert_allclose(trans[j], t, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(chk[j], t, rtol=1e-9, atol=1e-10)
            assert abs(lik[j] - l) <= 1e-12 * abs(l) and abs(chk_lik[j] - l) <= 1e-12 * abs(l)
    # the default table's three-way code on the R7.3 ACEGOT model
    pm7, om7 = _models(oracle, cases.MODEL_R73)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    job = cases.synthetic_jobs(cases.MODEL_R73, 1, 700, 50)[0]
    ref = list(job["ref"])
    for i in range(9, len(ref) - 6, 23):
        ref[i] = "L" if (i // 23) % 2 == 0 else "P"
    for i in (200, 201, 202):
        ref[i] = "L"
    job = dict(job, ref="".join(ref))
    trans, lik, _ = sa.expect_batch(pm7, p, [job])
    assert sa.expect_last_stats().n_ring_regions == 1
    t, l, _, _, _ = _oracle_expect(oracle, om7, job, op, ambig=oracle.ambig_map())
    np.testing.assert_allclose(trans[0], t, rtol=1e-9, atol=1e-10)
    assert abs(lik[0] - l) <= 1e-12 * abs(l)


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


def test_register_kernel_expectations_against_the_memory_resident_checker(oracle):
    """One path per cell: sa_expect_batch runs k_bwd_fast_expect (register sweeps; forward states of e-1 / e-2 in registers);
    SA_FLAG_FORCE_GENERIC keeps the memory-resident kernels, the checker.  Both against the oracle at 1e-9, on dense anchors
    (register sections), anchors thinned to a sixth (bands wider than a wave: the in-kernel memory-resident stretches), a read
    with several traceback segments and checkpoint groups, a tiny read, and no anchors at all."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 2, 900, 300) + cases.synthetic_jobs(cases.MODEL_6MER, 1, 3100, 310)
    jobs += cases.realistic_anchor_jobs(cases.MODEL_6MER, 1, 1200, 320)
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 1, 30, 330)
    bare = dict(jobs[0])
    bare["ax"], bare["ay"] = bare["ax"][:0], bare["ay"][:0]
    jobs.append(bare)
    fast_t, fast_l, _ = sa.expect_batch(pm, p, jobs)
    gen_t, gen_l, _ = sa.expect_batch(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC)
    for j, job in enumerate(jobs):
        t, l, _, _, _ = _oracle_expect(oracle, om, job, op)
        np.testing.assert_allclose(fast_t[j], t, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gen_t[j], t, rtol=1e-9, atol=1e-12)
        assert abs(fast_l[j] - l) <= 1e-12 * abs(l) and abs(gen_l[j] - l) <= 1e-12 * abs(l)
    # the register kernels really took these regions
    b = sa.Batch(pm, p, jobs[:3])
    assert b.stats().n_fast_regions == 3
    b.close()


def test_register_kernel_hdp_assignments_match_the_checker(oracle):
    """HDP: the assignment lists (transitions into the match state at or above the threshold, reference order) of the register
    kernels against the memory-resident kernels' and the oracle's, on the HDP workload's reads."""
    pm, om = _models(oracle, cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.05)
    op = cases.oracle_params(oracle, p)
    jobs = cases.hdp_jobs(2, 1100, 50, table5=pm.table5())
    ft, fl, fa = sa.expect_batch(pm, p, jobs)
    gt, gl, ga = sa.expect_batch(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC)
    for j, job in enumerate(jobs):
        t, l, pos, evs, st = _oracle_expect(oracle, om, job, op)
        np.testing.assert_allclose(ft[j], t, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(gt[j], t, rtol=1e-6, atol=1e-9)
        assert abs(fl[j] - l) <= 1e-9 * abs(l)
        assert len(pos) > 20
        a, g = [tuple(r) for r in fa[j].tolist()], [tuple(r) for r in ga[j].tolist()]
        if a != g:   # the two device logs differ in the last bit: only assignments sitting on the threshold may differ
            assert abs(len(a) - len(g)) <= 2 and len(set(a) ^ set(g)) <= 2
        mean = np.asarray(job["events"], dtype=np.float64)[:, 0]
        got = list(zip(fa[j][:, 0].tolist(), mean[fa[j][:, 1]].tolist()))
        exp = list(zip(pos.tolist(), evs.tolist()))
        if got != exp:
            assert abs(len(got) - len(exp)) <= 2 and len(set(got) ^ set(exp)) <= 2


def test_register_kernel_expectations_in_several_forward_storage_passes(oracle, monkeypatch):
    """The expectation batch keeps all three forward planes; with a small forward-storage budget it runs in several passes
    (regions packed into chunks) and must give the same numbers as in one."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 6, 800, 400)
    one_t, one_l, _ = sa.expect_batch(pm, p, jobs)
    monkeypatch.setenv("SA_F_BUDGET_CELLPATHS", "120000")
    b = sa.Batch(pm, p, jobs)
    assert b.stats().n_chunks >= 2
    b.close()
    many_t, many_l, _ = sa.expect_batch(pm, p, jobs)
    monkeypatch.delenv("SA_F_BUDGET_CELLPATHS")
    np.testing.assert_allclose(many_t, one_t, rtol=1e-12, atol=0)
    np.testing.assert_allclose(many_l, one_l, rtol=1e-14, atol=0)
    op = cases.oracle_params(oracle, p)
    t, l, _, _, _ = _oracle_expect(oracle, om, jobs[0], op)
    np.testing.assert_allclose(many_t[0], t, rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# The EM loops of the reference's own tests with the M-step in the LIBRARY (sa_hmm_*: create, add, normalize, load into
# the model) -- tests/stateMachineTests.c:1233-1330.  No oracle: the assertions are the reference's.
# ---------------------------------------------------------------------------------------------------------------------
def _zymo_em_inputs():
    import zymo_wholeread as z
    r = z.read_fixture()
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    ax, ay = z.remapped_anchors()
    b = z.BANDING
    p = sa.default_params(threshold=b["threshold"], expansion=b["expansion"], trace_back=b["trace_back"], min_diags=b["min_diags"],
                          split=b["split"])
    return z, r, alpha, k, t10, np.array(tab, dtype=np.float64), ax, ay, p


@pytest.mark.parametrize("emission", [0, 1])
def test_continuous_pair_hmm_em_on_the_zymo_read(emission):
    # test_continuousPairHmm_em (:1233-1281): loadDescaledStateMachine3, getExpectationsUsingAnchors(..., 0, 0) ten times, each
    # followed by continuousPairHmm_normalize and the two loads into the state machine; from the third iteration on
    # `pLikelihood <= likelihood * 0.85` (likelihoods are negative: the new one may not be more than ~18 % worse).  emission 1 is
    # the reference's (two distributions, descaled events), emission 0 the one signalMachine installs (default kernels).
    z, r, alpha, k, t10, tab, ax, ay, p = _zymo_em_inputs()
    tp = r["template_params"]
    t5 = tab.reshape(-1, 5).copy()                                   # emissions_signal_scaleNoise (impl/stateMachine.c:721-741)
    t5[:, 2] *= tp["scale_sd"]
    t5[:, 4] *= tp["var_sd"]
    t5[:, 3] = np.sqrt(np.power(t5[:, 2], 3.0) / t5[:, 4])
    m = sa.Model.create(alpha, k, t10, t5.reshape(-1))
    m.set_emission(emission)
    job = dict(ref=r["ref"], events=r["template_events"], ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"], var=tp["var"],
               ragged=(0, 0))
    prev, history = -np.inf, []
    for it in range(10):
        h = sa.Hmm.create(m, sa.HMM_GAUSSIAN, 0.0, 0.001, 0.001)     # continuousPairHmm_makeExpectationsHmm(sM, 0.001, 0.001)
        trans, lik, _ = sa.expect_batch(m, p, [job])
        h.add_expectations(trans[0], lik[0])
        assert np.all(h.transitions >= 0.001) and not h.observed.any()   # (the emission expectations are not updated: :914-944)
        h.normalize()
        h.load_into_model(m)
        if it > 1:
            assert prev <= h.likelihood * 0.85, (it, prev, h.likelihood)
        prev = h.likelihood
        history.append(prev)
        h.close()
    # ... and what EM promises, as far as this "likelihood" can show it: the figure is the reference's sum of the total
    # probability over ALL diagonals of every traceback (hmm->likelihood += totalProbability per diagonal,
    # impl/pairwiseAligner.c:1432), under transitions with a pseudocount -- it climbs steeply, then settles to within 1e-4
    assert all(b >= a - 1e-4 * abs(a) for a, b in zip(history, history[1:])), history
    assert history[-1] > history[0]
    t = m.transitions10()
    assert abs(t[0] + t[1] + t[2] - 1.0) < 1e-3 and abs(t[3] + t[4] - 1.0) < 1e-2 and abs(t[6] + t[8] - 1.0) < 1e-2


def test_hdp_hmm_em_transitions_on_the_zymo_read():
    # test_hdpHmm_emTransitions (:1283-1330): the bundled HDP, events descaled by nanopore_descaleNanoporeRead (the quirk of
    # zymo_wholeread.hdp_test_events), hdpHmm_makeExpectationsHmm(sM, p->threshold, 0.0), hmmDiscrete_normalizeTransitions and
    # continuousPairHmm_loadTransitionsIntoStateMachine ten times: `pLikelihood <= likelihood * 0.95` from the third iteration on
    z, r, alpha, k, t10, tab, ax, ay, p = _zymo_em_inputs()
    tp = r["template_params"]
    m = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    job = dict(ref=r["ref"], events=z.hdp_test_events(r), ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"], var=tp["var"],
               ragged=(0, 0))
    prev = -np.inf
    n_assign = []
    for it in range(10):
        h = sa.Hmm.create(m, sa.HMM_HDP, p.threshold, 0.0)
        trans, lik, assigns = sa.expect_batch(m, p, [job])
        h.add_expectations(trans[0], lik[0])
        for pos, ev in assigns[0]:
            h.add_assignment(job["ref"][int(pos):int(pos) + k], job["events"][int(ev), 0])
        n_assign.append(h.view().n_assignments)
        h.normalize()
        h.load_into_model(m)
        if it > 1:
            assert prev <= h.likelihood * 0.95, (it, prev, h.likelihood)
        prev = h.likelihood
        h.close()
    assert min(n_assign) > 500        # (to == match && p >= threshold: more than one assignment per second event)
