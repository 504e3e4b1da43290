"""Maximum-expected-accuracy path (src/signalalign/mea_algorithm.py:25-264) on the GPU, through the C ABI
(sa_mea_batch): the reference's known-answer matrix, bit-identical paths and sums against the CPU restatement on
random matrices and on posteriors produced by the HIP aligner itself, the reference's exception cases as status words,
and the global-front second pass."""
import json
import os

import numpy as np
import pytest
from scipy import sparse

import signalalign_amd as sa

import sa_cases as cases
from test_oracle_mea import random_prob_matrix

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _job(m, shortest):
    coo = sparse.coo_matrix(m)
    return dict(event_idx=coo.row, ref_idx=coo.col, posterior=coo.data, shortest=shortest)


def _check_against_oracle(oracle, jobs, got):
    for j, (job, (path, best, st, n_edges)) in enumerate(zip(jobs, got)):
        est, epath, ebest, esums = oracle.mea(job["event_idx"], job["ref_idx"], job["posterior"], job["shortest"],
                                              return_all=True)
        assert st == est, (j, st, est)
        if est == 0:
            assert np.array_equal(path, epath), j
            assert best == ebest, j                   # the same additions in the same order: equal, not close
            assert n_edges == len(esums), j
        else:
            assert len(path) == 0


def test_reference_known_answer_matrix():
    kat = json.load(open(os.path.join(HERE, "golden", "mea", "kat_5x5.json")))
    m = np.asarray(kat["matrix_ref_by_event"]).T
    jobs = [_job(m, c["shortest_ref_per_event"]) for c in kat["cases"]]
    got = sa.mea_batch(jobs)
    for (path, best, st, n_edges), c in zip(got, kat["cases"]):
        assert st == 0 and n_edges == c["n_edges"] and abs(best - 1.6) < 1e-7
        assert path.tolist() == [[0, 0], [1, 1], [1, 2], [3, 3], [4, 4]]


@pytest.mark.parametrize("tier", [0, 1, 2])
def test_random_matrices_bit_identical_to_the_oracle(oracle, monkeypatch, tier):
    # tier 0: fronts in registers (the default first stop), 1: in LDS, 2: in global memory -- the same answers
    monkeypatch.setenv("SA_MEA_TIER", str(tier))
    rng = np.random.default_rng(5)
    jobs = []
    for it in range(300):
        n_ev, n_ref = rng.integers(5, 60, 2)
        m, shortest = random_prob_matrix(rng, int(n_ev), int(n_ref), gaps=bool(it % 3))
        if np.count_nonzero(m.sum(axis=1)) < 2:
            continue
        jobs.append(_job(m, shortest))
    got = sa.mea_batch(jobs)
    _check_against_oracle(oracle, jobs, got)
    assert sum(1 for g in got if g[2] == 0) > 250


def test_exception_cases_become_status_words(oracle):
    e = np.zeros(0, dtype=np.int32)
    jobs = [dict(event_idx=e, ref_idx=e, posterior=np.zeros(0), shortest=[]),                     # ValueError
            dict(event_idx=[3, 3], ref_idx=[0, 1], posterior=[0.4, 0.6], shortest=[0, 0, 0, 0]),   # IndexError :61
            dict(event_idx=[0, 1], ref_idx=[0, 1], posterior=[0.4, 0.6], shortest=[0]),            # IndexError :106
            dict(event_idx=[0, 1], ref_idx=[0, 1], posterior=[0.4, 0.6], shortest=[0, 0])]         # fine
    got = sa.mea_batch(jobs)
    assert [g[2] for g in got] == [1, 2, 5, 0]
    assert got[3][0].tolist() == [[0, 0], [1, 1]] and got[3][1] == 1.0
    _check_against_oracle(oracle, jobs, got)
    assert sa.mea_batch([]) == []


def test_front_longer_than_the_lds_share_takes_the_global_pass(oracle):
    # a first event with 700 rising posteriors seeds a 700-edge front (registers hold 64, LDS 256): the read falls
    # through both faster tiers
    rng = np.random.default_rng(3)
    n_ref, n_ev = 700, 40
    m = np.zeros((n_ev, n_ref))
    m[0, :] = np.sort(rng.random(n_ref)) / n_ref
    for ev in range(1, n_ev - 1):
        cols = rng.choice(n_ref, 12, replace=False)
        m[ev, cols] = rng.random(12) / 12
    m[n_ev - 1, :] = np.sort(rng.random(n_ref)) / n_ref     # ... and a last event that leaves a long final front
    shortest = np.zeros(n_ev)
    small = _job(random_prob_matrix(rng, 20, 20)[0], np.zeros(20))
    jobs = [small, _job(m, shortest), small]
    got = sa.mea_batch(jobs)
    _check_against_oracle(oracle, jobs, got)
    assert got[1][2] == 0 and got[1][3] > 256        # the final front itself is longer than the LDS share


def test_front_between_64_and_256_edges_takes_the_lds_pass(oracle):
    rng = np.random.default_rng(4)
    n_ref, n_ev = 150, 30
    m = np.zeros((n_ev, n_ref))
    m[0, :] = np.sort(rng.random(n_ref)) / n_ref
    for ev in range(1, n_ev):
        cols = rng.choice(n_ref, 20, replace=False)
        m[ev, cols] = rng.random(20) / 20
    jobs = [_job(m, np.zeros(n_ev))]
    got = sa.mea_batch(jobs)
    _check_against_oracle(oracle, jobs, got)
    assert got[0][2] == 0


def test_posteriors_of_the_hip_aligner(oracle):
    """End to end as mea_alignment_from_signal_align (:323-341) chains it: aligned pairs -> event table columns ->
    sa_mea_params -> sa_mea_batch; the table holds the posterior the TSV prints (6 decimals)."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    reads = cases.synthetic_jobs(cases.MODEL_6MER, 6, 1500, 900)
    b = sa.Batch(pm, p, reads)
    b.run()
    jobs = []
    for j in range(len(reads)):
        pr = b.pairs(j)
        post = np.array([float("%f" % (q / 1e7)) for q in pr["prob_e7"]])
        ev, rf, po, sh = sa.mea_params(pr["x"] + 1000, pr["y"], post)
        oe = oracle.mea_params(pr["x"] + 1000, pr["y"], post)
        assert np.array_equal(ev, oe[0]) and np.array_equal(rf, oe[1]) and np.array_equal(po, oe[2]) and np.array_equal(sh, oe[3])
        jobs.append(dict(event_idx=ev, ref_idx=rf, posterior=po, shortest=sh))
    b.close()
    got = sa.mea_batch(jobs)
    _check_against_oracle(oracle, jobs, got)
    for (path, best, st, _), job in zip(got, jobs):
        assert st == 0
        assert np.all(np.diff(path[:, 0]) >= 0) and np.all(np.diff(path[:, 1]) > 0)
        # nearly every event of the read is on the path, and the path's expected accuracy is most of the mass
        assert len(path) > 0.9 * len(np.unique(job["event_idx"]))


@pytest.mark.parametrize("flags", [0, sa.FLAG_EXACT])
def test_chained_onto_a_finished_batch(oracle, flags):
    """sa_batch_mea builds the matrices on the device from the pairs sa_batch_run left there: the same paths as the
    explicit route (pairs to the host, "%f", get_mea_params_from_events, maximum_expected_accuracy_alignment) through
    the CPU restatement."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    reads = cases.synthetic_jobs(cases.MODEL_6MER, 8, 1200, 300)
    b = sa.Batch(pm, p, reads, flags=flags)
    b.run()
    got = b.mea()
    assert len(got) == len(reads)
    for j in range(len(reads)):
        pr = b.pairs(j)
        post = np.array([float("%f" % (q / 1e7)) for q in pr["prob_e7"]])
        ev, rf, po, sh = oracle.mea_params(pr["x"], pr["y"], post)
        est, epath, ebest = oracle.mea(ev, rf, po, sh)
        path, best, st = got[j]
        assert st == est == 0
        assert best == ebest
        x0, y0 = int(pr["x"].min()), int(pr["y"].min())
        assert np.array_equal(path[:, 0] - x0, epath[:, 0]) and np.array_equal(path[:, 1] - y0, epath[:, 1])
        # every pair of the path is one of the read's aligned pairs
        have = set(zip(pr["x"].tolist(), pr["y"].tolist()))
        assert all((int(x), int(y)) in have for x, y in path)
    b.close()


def test_chained_with_ambiguous_positions_and_an_empty_read(oracle):
    # several paths per cell (duplicate (x, y) rows with different posteriors): the lowest posterior of a cell stays
    pm = sa.Model.load(cases.MODEL_CPG)
    p = sa.default_params()
    reads = cases.synthetic_jobs(cases.MODEL_CPG, 3, 600, 40, cpg_ambiguous=True)
    b = sa.Batch(pm, p, reads, ambig=sa.default_ambig({"X": "CE"}))
    b.run()
    got = b.mea()
    for j in range(len(reads)):
        pr = b.pairs(j)
        post = np.array([float("%f" % (q / 1e7)) for q in pr["prob_e7"]])
        ev, rf, po, sh = oracle.mea_params(pr["x"], pr["y"], post)
        est, epath, ebest = oracle.mea(ev, rf, po, sh)
        path, best, st = got[j]
        assert st == est and best == ebest
        x0, y0 = int(pr["x"].min()), int(pr["y"].min())
        assert np.array_equal(path[:, 0] - x0, epath[:, 0]) and np.array_equal(path[:, 1] - y0, epath[:, 1])
    b.close()


def test_device_rounds_every_posterior_as_percent_f_does():
    """All 10^7 + 1 values of prob_e7: the GPU's printed posterior equals the host's (same source, the device's own
    division and fma), and the host's is checked against Python's "%f" in tests/test_host_mea.py."""
    import ctypes as C
    L = sa.lib()
    n = 10_000_001
    dev = np.zeros(n)
    assert L.sa_mea_printed_posterior_device(0, n, dev.ctypes.data_as(C.POINTER(C.c_double)), 0) == 0
    # the host formula, vectorised: k = prob // 10 rounded on the seventh decimal
    prob = np.arange(n, dtype=np.int64)
    k, rem = prob // 10, prob % 10
    up = rem > 5
    tie = rem == 5
    for v in prob[tie][:: 997]:                       # spot-check the vector formula's tie branch against the C one
        assert L.sa_mea_printed_posterior(int(v)) == dev[v]
    assert np.array_equal(dev[~tie], ((k + up)[~tie]) / 1e6)
    host_ties = np.array([L.sa_mea_printed_posterior(int(v)) for v in prob[tie][:: 13]])
    assert np.array_equal(dev[tie][:: 13], host_ties)


def test_chained_over_several_storage_passes_and_result_groups(oracle, monkeypatch):
    """The chained form finds each read's pairs where the result pipeline left them: several passes over the forward
    storage, several result groups per pass, a read without any pair in the middle."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    reads = cases.synthetic_jobs(cases.MODEL_6MER, 7, 900, 4100)
    reads.insert(3, dict(ref="ACGTACGT", events=np.zeros(0), ax=[], ay=[]))
    plain = sa.Batch(pm, p, reads)
    plain.run()
    want = plain.mea()
    plain.close()
    assert want[3][2] == 1 and len(want[3][0]) == 0          # SA_MEA_EMPTY for the read without pairs
    for env in ({"SA_F_BUDGET_CELLPATHS": "200000"}, {"SA_GROUPS": "3"}, {"SA_GROUPS": "2", "SA_F_BUDGET_CELLPATHS": "200000"}):
        monkeypatch.delenv("SA_F_BUDGET_CELLPATHS", raising=False)
        monkeypatch.delenv("SA_GROUPS", raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        b = sa.Batch(pm, p, reads)
        for _ in range(2):                                   # the second run takes the overlapped-copy path
            b.run()
            got = b.mea()
            for (gp, gs, gst), (wp, ws, wst) in zip(got, want):
                assert gst == wst and gs == ws and np.array_equal(gp, wp), env
        b.close()


@pytest.mark.parametrize("tier", [0, 1, 2])
def test_ties_everywhere(oracle, monkeypatch, tier):
    """Posteriors quantised to eighths: equal sums and equal posteriors are the rule, so every strict / non-strict
    comparison of the reference decides something (first event kept while not falling, 'raises the maximum' strictly,
    first best edge wins).  The register tier's ballots rely on sums never falling along a front; ties are where that
    would show."""
    monkeypatch.setenv("SA_MEA_TIER", str(tier))
    rng = np.random.default_rng(99)
    jobs = []
    for it in range(400):
        n_ev, n_ref = rng.integers(3, 40, 2)
        m = rng.integers(0, 9, (int(n_ev), int(n_ref))) / 8.0
        m[rng.random(m.shape) < 0.6] = 0.0
        rows = np.nonzero(m.sum(axis=1))[0]
        if len(rows) < 2:
            continue
        # shortest_ref_per_event as the reference derives it: lowest column of this and all later events
        shortest = np.full(int(n_ev), np.inf)
        low = np.inf
        for e in range(int(n_ev) - 1, -1, -1):
            nz = np.nonzero(m[e])[0]
            if len(nz):
                low = min(low, nz[0])
                shortest[e] = low
        jobs.append(_job(m, shortest))
    got = sa.mea_batch(jobs)
    _check_against_oracle(oracle, jobs, got)
    assert sum(1 for g in got if g[2] == 0) > 300
