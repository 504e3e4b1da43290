"""Properties the reference's whole-read tests assert (tests/stateMachineTests.c), restated on inputs available here.

  * test_DegenerateNucleotides :920-983 -- an ambiguous letter with a single substitution option gives the same result
    as writing the base itself (1076 / 1076 / 1076 pairs for C, C->E, C->O in the reference's fixture).
  * test_stateMachine3_getAlignedPairsWithBanding :842-872 -- banded == un-banded.
  * test_continuousPairHmm_em :1233-1283 -- the likelihood does not decrease over EM iterations on the transitions
    (expectations -> row normalisation -> continuousPairHmm_loadTransitionsIntoStateMachine).
"""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def _run(pm, p, jobs, flags=0, ambig=None):
    b = sa.Batch(pm, p, jobs, ambig=ambig, flags=flags)
    b.run()
    out = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    return out


def test_single_option_ambiguity_equals_the_base():
    pm = sa.Model.load(cases.MODEL_CPG)            # ACEGT
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, 3, 700, 60)
    plain = _run(pm, p, jobs, flags=sa.FLAG_EXACT)
    for letter, base in (("X", "C"), ("X", "E")):
        swapped = []
        for job in jobs:
            j2 = dict(job)
            j2["ref"] = job["ref"].replace("C", letter)   # every C becomes the ambiguous letter ...
            swapped.append(j2)
        got = _run(pm, p, swapped, flags=sa.FLAG_EXACT, ambig=sa.default_ambig({letter: base}))  # ... with one option
        if base == "C":
            for a, b in zip(got, plain):
                assert np.array_equal(a, b)            # bit-identical, k-mer ids included
        else:
            # C -> E reads a different k-mer table row: same coordinates cannot be demanded, the pair count stays close
            for a, b in zip(got, plain):
                assert 0.5 * len(b) < len(a) < 2 * len(b)
        # and the register kernels (one path per cell) take these regions: the expansion produced no extra paths
        b = sa.Batch(pm, p, swapped, ambig=sa.default_ambig({letter: base}))
        assert b.stats().n_fast_regions == len(swapped)
        b.close()


def test_banded_equals_unbanded():
    pm = sa.Model.load(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 500, 900)
    narrow = _run(pm, sa.default_params(expansion=50), jobs)
    wide = _run(pm, sa.default_params(expansion=2000), jobs)   # the band covers the whole matrix
    for a, b in zip(narrow, wide):
        ka = {(int(r["x"]), int(r["y"])): int(r["prob_e7"]) for r in a}
        kb = {(int(r["x"]), int(r["y"])): int(r["prob_e7"]) for r in b}
        common = set(ka) & set(kb)
        assert len(common) >= 0.995 * max(len(ka), len(kb))
        assert max(abs(ka[c] - kb[c]) for c in common) <= 2000   # what the band cuts off is worth < 2e-4 of a posterior


def test_em_on_transitions_does_not_lose_likelihood():
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 6, 800, 40)
    p = sa.default_params()
    t = np.array(t10, dtype=np.float64)
    # start from a deliberately poor transition matrix
    t[[0, 1, 2]] = [0.4, 0.3, 0.3]
    t[[3, 4]] = [0.5, 0.5]
    t[[6, 8]] = [0.5, 0.5]
    history = []
    pm = sa.Model.create(alpha, k, t, tab)
    for it in range(6):
        # the library's expectations object and M-step (sa_hmm_*: hmmContinuous_getExpectationsHmm, continuousPairHmm_normalize,
        # continuousPairHmm_loadTransitionsIntoStateMachine, impl/continuousHmm.c:282-351)
        h = sa.Hmm.create(pm, sa.HMM_GAUSSIAN, 0.0, 0.001, 0.001)
        trans, lik, _ = sa.expect_batch(pm, p, jobs)
        for j in range(len(jobs)):
            h.add_expectations(trans[j], lik[j])
        history.append(h.likelihood)
        h.normalize()
        h.load_into_model(pm)
        h.close()
    pm.close()
    assert all(b >= a - 1e-6 * abs(a) for a, b in zip(history, history[1:])), history
    assert history[-1] > history[0]
