"""Host side of the maximum-expected-accuracy step: sa_mea_params (get_mea_params_from_events,
src/signalalign/mea_algorithm.py:267-320) against the CPU restatement.  No GPU needed."""
import numpy as np

import signalalign_amd as sa


def test_params_match_the_oracle_on_random_tables(oracle):
    rng = np.random.default_rng(11)
    for it in range(40):
        n = int(rng.integers(1, 400))
        ev = rng.integers(100, 100 + max(2, n // 3), n)
        ref = rng.integers(5000, 5000 + max(2, n // 4), n)
        post = np.round(rng.random(n), 2)            # coarse values: duplicates of a cell and exact zeros both occur
        if it % 2:
            ref = ref.max() - ref + 17               # minus-strand looking tables too
        g = sa.mea_params(ref, ev, post)
        e = oracle.mea_params(ref, ev, post)
        for a, b in zip(g, e):
            assert np.array_equal(a, b)


def test_params_edge_cases(oracle):
    g = sa.mea_params([5, 6, 7, 7, 6, 9], [0, 0, 1, 1, 1, 3], [0.5, 0.5, 0.9, 0.3, 0.0, 1.0])
    assert g[0].tolist() == [0, 0, 1, 3] and g[1].tolist() == [0, 1, 2, 4] and g[2].tolist() == [0.5, 0.5, 0.3, 1.0]
    assert g[3].tolist() == [0, 1, sa.MEA_INF, 4]
    one = sa.mea_params([42], [7], [0.25])
    assert one[0].tolist() == [0] and one[1].tolist() == [0] and one[3].tolist() == [0]


def test_printed_posterior_is_what_percent_f_prints():
    """sa_mea_printed_posterior(prob_e7) == float("%f" % (prob_e7 / 1e7)): the device builds the event table's
    posterior column with this formula.  Every value whose seventh decimal is a 5 (the only candidates for a tie) in a
    stride, plus random others, against Python's own formatting."""
    L = sa.lib()
    rng = np.random.default_rng(1)
    values = list(range(5, 10_000_000, 10 * 97)) + list(range(0, 2000)) + [9_999_995, 10_000_000, 78125, 234375] \
        + rng.integers(0, 10_000_001, 20000).tolist()
    for v in values:
        assert L.sa_mea_printed_posterior(int(v)) == float("%f" % (v / 1e7)), v
