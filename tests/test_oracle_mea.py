"""Pins the CPU restatement of the maximum-expected-accuracy step (oracle/sa_mea_oracle.c) the way the reference's own
tests do (src/signalalign/tests/test_mea_algorithm.py): its 5x5 known-answer matrix, agreement with an independent
exhaustive formulation on random matrices, the traceback invariants, and the event-table -> matrix conversion."""
import json
import os

import numpy as np
import pytest
from scipy import sparse

from oracle import sa_oracle_py as oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def random_prob_matrix(rng, n_events, n_refs, gaps=True):
    """Random test input shaped like the reference's generator (mea_algorithm.py:511-553): every event row holds a
    random number of normalised probabilities on shuffled reference columns; returns the matrix and, per event, the
    smallest reference column of this and all later events."""
    m = np.zeros((n_events, n_refs))
    shortest = np.zeros(n_events)
    lowest = n_refs
    start = int(rng.integers(0, 3)) if gaps else 0
    cols = list(range(n_refs))
    for ev in range(n_events - 1, start - 1, -1):
        probs = rng.random(int(rng.integers(0 if gaps else 1, n_refs)))
        if len(probs):
            probs = np.sort(probs / probs.sum())
        rng.shuffle(cols)
        if not gaps and ev == n_events - 1:
            cols.remove(n_refs - 1)
            cols.insert(0, n_refs - 1)
        if not gaps and ev == 0:
            cols.remove(0)
            cols.insert(0, 0)
        for p, c in zip(probs, cols):
            m[ev, c] = p
            lowest = min(lowest, c)
        shortest[ev] = lowest
    return m, shortest


def test_reference_known_answer_matrix():
    kat = json.load(open(os.path.join(HERE, "golden", "mea", "kat_5x5.json")))
    m = np.asarray(kat["matrix_ref_by_event"]).T          # events x reference
    coo = sparse.coo_matrix(m)
    for case in kat["cases"]:
        st, path, best, sums = oracle.mea(coo.row, coo.col, coo.data, case["shortest_ref_per_event"], return_all=True)
        assert st == 0
        assert len(sums) == case["n_edges"]
        if case["edge_sums"]:
            assert np.allclose(sums, case["edge_sums"], atol=1e-7)       # assertAlmostEqual, 7 places
    # "0.2->0.5->0.1->0.4->0.5 = 1.6 (don't count the horizontal move from 0.5 to 0.1)"
    st, path, best = oracle.mea(coo.row, coo.col, coo.data, [0, 0, 0, 3, 3])
    assert path.tolist() == [[0, 0], [1, 1], [1, 2], [3, 3], [4, 4]] and abs(best - 1.6) < 1e-12


def test_agrees_with_exhaustive_formulation_and_traceback_invariants():
    rng = np.random.default_rng(20180125)
    for _ in range(60):                                    # test_mae_random_matrix: 20 draws of 20..40 squares
        n_ev, n_ref = rng.integers(20, 40, 2)
        m, shortest = random_prob_matrix(rng, int(n_ev), int(n_ref))
        coo = sparse.coo_matrix(m)
        st, path, best = oracle.mea(coo.row, coo.col, coo.data, shortest)
        assert st == 0
        assert abs(best - oracle.mea_exhaustive(m, shortest)) < 1e-7
        # walk the path as the reference's test does: never back along the reference, one step per event, and the sum
        # counts a posterior only where the reference position changes (plus the first one)
        assert np.all(np.diff(path[:, 0]) >= 0) and np.all(np.diff(path[:, 1]) > 0)
        total = m[path[0, 1], path[0, 0]]
        for q in range(1, len(path)):
            if path[q, 0] != path[q - 1, 0]:
                total += m[path[q, 1], path[q, 0]]
        assert abs(total - best) < 1e-7


def test_error_statuses_mirror_the_reference_exceptions():
    assert oracle.mea([], [], [], [])[0] == 1                                   # min() of an empty sequence
    assert oracle.mea([3, 3], [0, 1], [0.4, 0.6], [0, 0, 0, 0])[0] == 2         # one event only: IndexError
    assert oracle.mea([0, 1], [0, 1], [0.4, 0.6], [0])[0] == 5                  # event outside shortest_ref_per_event


def test_event_table_to_sparse_matrix():
    """test_get_mea_params_from_events: a table generated from a matrix converts back to the same matrix and
    shortest_ref_per_event; also on the minus strand (descending reference indices) and with shuffled rows."""
    rng = np.random.default_rng(7)

    def check(m, shortest, minus):
        n_ref = m.shape[1]
        ev, ref = np.nonzero(m)
        post = m[ev, ref]
        order = rng.permutation(len(ev))
        ref_col = (1000 + (n_ref - 1 - ref)) if minus else (1000 + ref)
        rows, cols, data, sh = oracle.mea_params(ref_col[order], 50 + ev[order], post[order])
        coo = sparse.coo_matrix(m)
        assert rows.tolist() == coo.row.tolist() and cols.tolist() == coo.col.tolist()
        assert data.tolist() == coo.data.tolist()
        assert sh.tolist() == np.asarray(shortest).astype(int).tolist()

    for it in range(20):
        n_ev, n_ref = rng.integers(10, 20, 2)
        m, shortest = random_prob_matrix(rng, int(n_ev), int(n_ref), gaps=False)
        check(m, shortest, minus=False)
    # minus strand: the table's reference indices fall as the events advance (:288-303 flips them); an alignment-like
    # band, so that the reference's test "first row above last row" can tell
    for it in range(10):
        n_ev = int(rng.integers(10, 30))
        m = np.zeros((n_ev, n_ev + 3))
        for e in range(n_ev):
            w = rng.random(3)
            m[e, e:e + 3] = w / w.sum()
        shortest = np.arange(n_ev)
        check(m, shortest, minus=True)
        check(m, shortest, minus=False)


def test_event_table_duplicates_zeros_and_missing_events():
    # event 2 has no rows (inf), the cell (event 1, ref 2) appears twice (numpy's field-order tie break puts the lower
    # posterior first, and the first row is the one that stays), a zero posterior counts for shortest_ref only
    ref = np.array([5, 6, 7, 7, 6, 9])
    ev = np.array([0, 0, 1, 1, 1, 3])
    post = np.array([0.5, 0.5, 0.9, 0.3, 0.0, 1.0])
    rows, cols, data, sh = oracle.mea_params(ref, ev, post)
    assert rows.tolist() == [0, 0, 1, 3] and cols.tolist() == [0, 1, 2, 4] and data.tolist() == [0.5, 0.5, 0.3, 1.0]
    assert sh.tolist() == [0, 1, oracle.MEA_INF, 4]
