"""The expectations objects (sa_hmm_*: Hmm / ContinuousPairHmm / HdpHmm of impl/continuousHmm.c), host only.

Mirrors of the reference's own tests, with their literal numbers (paths relative to the upstream tree):
  tests/stateMachineTests.c:998-1018   test_makeAndCheckModels
  tests/stateMachineTests.c:1020-1155  test_continuousPairHmm          (write -> read -> equal, accumulators, normalize)
  tests/stateMachineTests.c:1157-1231  test_hdpHmmWithoutAssignments   (write -> read -> equal with three assignments)
and the reference's golden .expectations file (tests/test_expectation_files/4f9a316c-...template.expectations.tsv, committed as
data under tests/golden/format) read by sa_hmm_load and written back by sa_hmm_write byte for byte.
The EM loops of :1233-1330 need the GPU: tests/test_gpu_expectations.py.
"""
import gzip
import os

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases


def _model():
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_R73)
    return sa.Model.create(alpha, k, t10, tab), np.array(tab).reshape(-1, 5)


def test_make_and_check_models():
    m, tab = _model()
    h = sa.Hmm.create(m, sa.HMM_GAUSSIAN, 0.0, 0.0, 0.0)
    v = h.view()
    assert v.has_model == 1 and v.n_states == 3 and v.n_kmers == 6 ** 6 and v.alphabet.decode() == "ACEGOT"
    assert np.array_equal(h.event_model, tab)          # test_cpHmmEmissionsAgainstStateMachine: equal to the last bit
    assert np.all(h.transitions == 0.0) and h.likelihood == 0.0


def test_continuous_pair_hmm_round_trip_and_normalize(tmp_path):
    m, tab = _model()
    h = sa.Hmm.create(m, sa.HMM_GAUSSIAN, 0.0, 0.0, 0.0)
    n = 3
    h.add_expectations(np.arange(9, dtype=np.float64), 0.0)             # dummy = from * nStates + to
    path = str(tmp_path / "temp.hmm")
    h.write(path)
    h.close()
    h = sa.Hmm.load(path, sa.HMM_GAUSSIAN, 0.0, 0.0)
    assert np.array_equal(h.transitions, np.arange(9, dtype=np.float64).reshape(3, 3))
    # the event model went through "%lf": level mean and sd (six decimals in the model file) come back exactly -- what
    # test_cpHmmEmissionsAgainstStateMachine asserts with a tolerance of 0.0 -- the longer noise columns to the printed digit
    assert np.array_equal(h.event_model[:, :2], tab[:, :2]) and np.abs(h.event_model - tab).max() <= 5e-7
    tab_file = h.event_model.copy()
    means = h.event_model[:, 0].copy()
    for i in range(len(means)):
        h.add_emission_expectation(i, means[i], 1)
    assert np.array_equal(h.event_expectations[:, 0], means) and np.all(h.event_expectations[:, 1] == 0.0)
    assert np.all(h.posteriors == 1.0)
    for i in range(len(means)):
        h.add_emission_expectation(i, 2 * means[i], 1)
    assert np.all(h.posteriors == 2.0)
    assert np.array_equal(h.event_expectations[:, 0], means * 3)
    assert np.array_equal(h.event_expectations[:, 1], (2 * means - 1.5 * means) ** 2)
    assert np.all(h.observed == 1)
    h.normalize()
    assert np.all(h.posteriors == 2.0)
    for frm in range(n):
        z = frm * n * n + (n * (n - 1)) / 2
        for to in range(n):
            assert h.transitions[frm, to] == (frm * n + to) / z
    assert np.array_equal(h.event_model[:, 0], 1.5 * tab[:, 0])
    assert np.array_equal(h.event_model[:, 1], np.sqrt((2 * tab[:, 0] - 1.5 * tab[:, 0]) ** 2 / 2))
    assert np.array_equal(h.event_model[:, 2:], tab_file[:, 2:])
    # the M-step: transitions (log) and level mean / sd into the model, noise columns untouched
    h.load_into_model(m)
    t10 = m.transitions10()
    T = h.transitions
    np.testing.assert_allclose(t10[[0, 1, 2, 3, 4, 6, 8]], [T[0, 0], T[0, 1], T[0, 2], T[1, 0], T[1, 1], T[2, 0], T[2, 2]], rtol=1e-15)
    assert t10[5] == 0.0 and t10[7] == 0.0          # gapX <-> gapY stay dead (sa_hmm.c)
    t5 = np.array(m.table5()).reshape(-1, 5)
    assert np.array_equal(t5[:, :2], h.event_model[:, :2]) and np.array_equal(t5[:, 2:], tab[:, 2:])


def test_hdp_hmm_with_assignments_round_trip(tmp_path):
    m, tab = _model()
    h = sa.Hmm.create(m, sa.HMM_HDP, 0.0, 0.0)
    h.add_expectations(np.arange(9, dtype=np.float64), 0.0)
    sequence, fake = "ACGTCATACATGACTATA", [65.0, 64.0, 63.0]
    for a in range(3):
        h.add_assignment(sequence[a * 6:], fake[a])        # a pointer into the sequence: the k-mer starts there
    assert h.view().n_assignments == 3
    path = str(tmp_path / "temp_hdp.hmm")
    h.write(path)
    lines = open(path).read().split("\n")
    assert lines[3] == "65.000000\t64.000000\t63.000000\t" and lines[4] == "ACGTCA\tTACATG\tACTATA\t"
    h.close()
    h = sa.Hmm.load(path, sa.HMM_HDP)
    assert np.array_equal(h.transitions, np.arange(9, dtype=np.float64).reshape(3, 3))
    kmers, events = h.assignments()
    assert kmers == ["ACGTCA", "TACATG", "ACTATA"] and events.tolist() == fake
    h.normalize()
    for frm in range(3):
        z = frm * 9 + 3
        for to in range(3):
            assert h.transitions[frm, to] == (frm * 3 + to) / z
    np.testing.assert_allclose(h.event_model, tab, atol=1e-4, rtol=0)
    with pytest.raises(sa.SaError):
        h.add_emission_expectation(0, 1.0, 1.0)            # an HdpHmm has no emission expectations


def test_reference_golden_expectations_file_goes_through_load_and_write(tmp_path):
    gz = os.path.join(cases.GOLDEN, "format", "4f9a316c-8bb3-410a-8cfc-026061f7e8db.template.expectations.tsv.gz")
    text = gzip.open(gz, "rt").read()
    src = str(tmp_path / "golden.expectations.tsv")
    open(src, "w").write(text)
    # hmmContinuous_loadSignalHmmFromFile(hmmFile, type, 0.0, 0.001) (impl/continuousHmm.c:805): the pseudocounts of a load
    h = sa.Hmm.load(src, sa.HMM_GAUSSIAN, 0.0, 0.001)
    v = h.view()
    assert (v.n_alpha, v.k, v.n_kmers, v.alphabet.decode()) == (5, 6, 15625, "ACEGT")
    toks = text.split("\n")[1].split()
    assert np.array_equal(h.transitions.reshape(-1), np.array(toks[:9], dtype=np.float64)) and h.likelihood == float(toks[9])
    assert h.likelihood < -1e8                              # the per-diagonal sum
    # the reference's reader stops behind the event model (impl/continuousHmm.c:409-507): a loaded object starts with empty
    # accumulators at the pseudocounts, whatever the file's last three lines hold
    assert np.all(h.event_expectations == 0.0) and np.all(h.posteriors == 0.001) and not h.observed.any()
    # trainModels.py's reader (HMM.add_expectations_file) takes all six lines: added to an empty object and written back, the
    # reference's file comes out byte for byte
    acc = sa.Hmm.load(src, sa.HMM_GAUSSIAN, 0.0, 0.0)
    acc.transitions[:] = 0.0
    acc.view().likelihood[0] = 0.0
    acc.add_expectations_file(src)
    assert acc.observed.sum() > 1000 and acc.posteriors.max() > 1.0
    out = str(tmp_path / "back.expectations.tsv")
    acc.write(out)
    assert open(out).read() == text
    acc.add_expectations_file(src)                          # a second read's file of the same content: everything doubles
    assert np.array_equal(acc.transitions.reshape(-1), 2 * np.array(toks[:9], dtype=np.float64))
    empty = str(tmp_path / "empty.tsv")
    open(empty, "w").close()
    with pytest.raises(sa.SaError):
        acc.add_expectations_file(empty)
    # and the file's transitions normalise to a stochastic matrix that goes into a model of the same shape
    h.normalize()
    assert np.allclose(h.transitions.sum(axis=1), 1.0)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_CPG)
    m = sa.Model.create(alpha, k, t10, tab)
    h.load_into_model(m)
    np.testing.assert_allclose(m.transitions10()[[0, 1, 2]], h.transitions[0], rtol=1e-15)
    m5 = sa.Model.load(cases.MODEL_5MER)
    with pytest.raises(sa.SaError):
        h.load_into_model(m5)                               # another alphabet / k-mer length


def test_malformed_expectations_files_are_refused(tmp_path):
    m, tab = _model()
    h = sa.Hmm.create(m, sa.HMM_GAUSSIAN, 0.0, 0.001, 0.001)
    good = str(tmp_path / "good.hmm")
    h.write(good)
    lines = open(good).read().split("\n")
    cases_ = {
        "header": ["3\t6\tACEGOT\t"] + lines[1:],                               # three tokens
        "states": ["5\t6\tACEGOT\t6\t"] + lines[1:],
        "alphabet": ["3\t5\tACEGOT\t6\t"] + lines[1:],
        "transitions": [lines[0], "\t".join(lines[1].split("\t")[:9])] + lines[2:],   # the likelihood is missing
        "number": [lines[0], lines[1].replace("0.001000", "abc", 1)] + lines[2:],
        "model": lines[:2] + ["\t".join(lines[2].split("\t")[:-2])] + lines[3:],
        "eof": lines[:2],
    }
    for name, ls in cases_.items():
        bad = str(tmp_path / (name + ".hmm"))
        open(bad, "w").write("\n".join(ls))
        with pytest.raises(sa.SaError) as ei:
            sa.Hmm.load(bad, sa.HMM_GAUSSIAN)
        assert ei.value.code == -6, name
    with pytest.raises(sa.SaError):
        sa.Hmm.load(str(tmp_path / "missing.hmm"))
    # a NaN transition leaves an empty file behind (hmmContinuous_checkTransitions)
    h.transitions[1, 1] = float("nan")
    nanp = str(tmp_path / "nan.hmm")
    h.write(nanp)
    assert os.path.getsize(nanp) == 0
    # an HDP file whose two assignment lines disagree
    hd = sa.Hmm.create(m, sa.HMM_HDP, 0.01, 0.0)
    hd.add_assignment("ACGTCA", 60.0)
    p = str(tmp_path / "hdp.hmm")
    hd.write(p)
    ls = open(p).read().split("\n")
    ls[3] = "60.000000\t61.000000\t"
    open(p, "w").write("\n".join(ls))
    with pytest.raises(sa.SaError):
        sa.Hmm.load(p, sa.HMM_HDP)
