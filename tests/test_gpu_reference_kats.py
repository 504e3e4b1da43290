"""The reference's OWN known answers, computed by libsignalalign_hip.so on the GPU -- no oracle anywhere in this file.

What the numbers are (paths relative to the upstream signalAlign tree):

* tests/stateMachineTests.c:441-565 / :567-698  two literal 7-event matrices, getStateMachine3 (the two-distribution emission
  without descaling, impl/stateMachine.c:659-700, :1757-1765), no anchors, band expansion 2, start / end state vectors NOT ragged,
  threshold 0.2: exactly 14 and 7 aligned pairs, every one inside the listed (x, y) set.
* tests/stateMachineTests.c:842-852  ZymoC_ch_1_file1.npRead x ZymoRef.txt, anchors from the reference's lastz run,
  getAlignedPairsUsingAnchors(..., 1, 1) with the scaled model (emissions_signal_scaleModel) and with the descaled,
  noise-scaled one (getStateMachine3_descaled(..., TRUE)): exactly 1076 pairs each.
* tests/stateMachineTests.c:920-983  the same read with every C of the reference replaced by C / E / O / the three-way code L,
  getAlignedPairsUsingAnchors(..., 0, 0) (inc/pairwiseAligner.h:406-414: both ends NOT ragged): 1076 / 1076 / 1076 / 7349.

The ragged-end arguments are sa_job_t.ends (include/signalalign_hip.h); the emissions are SA_EMISSION_TWO_DIST (descaled events)
and SA_EMISSION_TWO_DIST_SCALED_MODEL (scaled model).  Round 6: with flags 0 a batch whose regions all hold one path per cell runs
these emissions on the REGISTER kernels (k_fwd_fast_two / k_bwd_fast_two -- the kernels the bench times, with the noise term added:
wide stretches through their in-kernel memory-resident path), so the 14 / 7 matrices and the whole-read 1076 below come out of that
kernel family itself; SA_FLAG_EXACT, and any batch with several paths per cell (the degenerate-nucleotide test), takes the
reference-ordered memory-resident kernels.  The strip / ring kernels carry the emission signalMachine installs (MeanOnly) and meet
non-ragged ends in tests/test_gpu_fuzz.py and tests/test_gpu_parity.py against the CPU restatement.  Inputs: the .npRead / reference / model files the reference ships (tests/golden), the committed cigar lines
of its lastz subprocess (tests/zymo_wholeread.py restates what follows the subprocess in plain Python).
"""
import os

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import zymo_wholeread as z

pytestmark = pytest.mark.gpu

GOLDEN = z.GOLDEN
EM_TWO_DIST, EM_TWO_DIST_SCALED_MODEL = 1, 2

# the literal event records of the two tests (mean, noise, duration, start): tests/stateMachineTests.c:444-453, :570-579
SY6 = [58.743435, 0.887833, 0.0571, 0.0,
       53.604965, 0.816836, 0.0571, 0.1,
       58.432015, 0.735143, 0.0571, 0.2,
       63.684352, 0.795437, 0.0571, 0.3,
       58.921430, 0.812959, 0.0571, 0.4,
       59.895882, 0.740952, 0.0571, 0.5,
       61.684303, 0.722332, 0.0571, 0.67]
SY5 = [70.0423375640843, 2.1070814631739, 0.0571, 0.0,
       73.7087073662952, 1.90162684687837, 0.0571, 0.1,
       105.375581864011, 2.87252862011704, 0.0571, 0.2,
       82.9620934477158, 2.38320603353748, 0.0571, 0.3,
       84.6977645711335, 3.08486975249442, 0.0571, 0.4,
       58.0551144225027, 2.52297561817531, 0.0571, 0.5,
       94.337668063878, 1.9731952395105, 0.0571, 0.67]


def _align(model, params, job, flags):
    b = sa.Batch(model, params, [job], flags=flags)
    b.run()
    got = b.pairs(0)
    st = b.stats()
    b.close()
    return got, st


@pytest.mark.parametrize("flags", [0, sa.FLAG_EXACT])
@pytest.mark.parametrize("model_file,ref,sy,n_pairs,allowed", [
    ("testModelR73_acegot_template.model", "ACGATALGGACAT", SY6, 14,
     {(0, 0), (1, 1), (2, 2), (3, 3), (4, 3), (5, 4), (6, 5), (7, 6)}),
    ("testModelR9_5mer_acgt_template.model", "ACGATATGGACAT", SY5, 7,
     {(0, 0), (1, 1), (2, 2), (3, 3), (5, 4), (6, 5), (8, 6)}),
])
def test_literal_seven_event_matrices(model_file, ref, sy, n_pairs, allowed, flags):
    # test_sm3_diagonalDPCalculations / test_sm3_5merDiagonalDPCalculations: band_construct(no anchors, expansion 2),
    # startStateProb on diagonal 0 and endStateProb on the last one (not the ragged vectors), threshold 0.2
    alpha, k, t10, tab = synth.parse_model_table(os.path.join(GOLDEN, "models", model_file))
    m = sa.Model.create(alpha, k, t10, tab)
    m.set_emission(EM_TWO_DIST_SCALED_MODEL)
    p = sa.default_params(threshold=0.2, expansion=2, trace_back=40)
    job = dict(ref=ref, events=np.array(sy, dtype=np.float64).reshape(7, 4), ax=[], ay=[], ragged=(0, 0))
    got, st = _align(m, p, job, flags)
    if flags == 0 and "L" not in ref:
        assert st.n_fast_regions == st.n_regions >= 1 and st.n_ring_regions == 0   # the register kernels (round 6)
    else:   # (the first matrix holds the three-way code L: several paths per cell)
        assert st.n_fast_regions == 0 and st.n_ring_regions == 0                   # the reference-ordered kernels
    assert len(got) == n_pairs
    assert {(int(q["x"]), int(q["y"])) for q in got} <= allowed
    assert got["prob_e7"].min() >= 2000000 and got["prob_e7"].max() <= 10000000
    m.close()


def _zymo():
    r = z.read_fixture()
    alpha, k, t10, tab = synth.parse_model_table(os.path.join(GOLDEN, "models", "testModelR73_acegot_template.model"))
    ax, ay = z.remapped_anchors()
    b = z.BANDING
    p = sa.default_params(threshold=b["threshold"], expansion=b["expansion"], trace_back=b["trace_back"], min_diags=b["min_diags"],
                          split=b["split"])
    return r, alpha, k, t10, tab, ax, ay, p


def _noise_scaled(table5, tp):
    """emissions_signal_scaleNoise (impl/stateMachine.c:721-741) on the five columns of the model table"""
    t = np.array(table5, dtype=np.float64).reshape(-1, 5).copy()
    t[:, 2] = t[:, 2] * tp["scale_sd"]
    t[:, 4] = t[:, 4] * tp["var_sd"]
    t[:, 3] = np.sqrt(np.power(t[:, 2], 3.0) / t[:, 4])
    return t.reshape(-1)


def _check_pairs(got, lX, lY, unique):
    # checkAlignedPairs / checkAlignedPairsWithOverlap (tests/stateMachineTests.c:154-221)
    assert got["x"].min() >= 0 and got["x"].max() < lX and got["y"].min() >= 0 and got["y"].max() < lY
    assert got["prob_e7"].min() > 0 and got["prob_e7"].max() <= 10000000
    if unique:
        assert len({(int(a), int(b)) for a, b in zip(got["x"], got["y"])}) == len(got)


@pytest.mark.parametrize("flags", [0, sa.FLAG_EXACT])
def test_zymo_whole_read_banded_1076_scaled_and_descaled_model(flags):
    # test_stateMachine3_getAlignedPairsWithBanding -> test_stateMachine (:823-840): ragged 1, 1
    r, alpha, k, t10, tab, ax, ay, p = _zymo()
    tp = r["template_params"]
    assert len(ax) == 39
    lX, lY = len(r["ref"]) - (k - 1), r["template_events"].shape[0]
    scaled = sa.Model.create(alpha, k, t10, z.scaled_table(tab, tp))            # loadScaledStateMachine3 (:69-79)
    scaled.set_emission(EM_TWO_DIST_SCALED_MODEL)
    got, st = _align(scaled, p, dict(ref=r["ref"], events=r["template_events"], ax=ax, ay=ay, ragged=(1, 1)), flags)
    assert (st.n_fast_regions == st.n_regions >= 1) if flags == 0 else (st.n_fast_regions == 0)
    assert len(got) == z.N_PAIRS_GAUSS
    _check_pairs(got, lX, lY, True)
    descaled = sa.Model.create(alpha, k, t10, _noise_scaled(tab, tp))           # loadDescaledStateMachine3 (:81-88)
    descaled.set_emission(EM_TWO_DIST)
    job = dict(ref=r["ref"], events=r["template_events"], ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"], var=tp["var"],
               ragged=(1, 1))
    got, st = _align(descaled, p, job, flags)
    assert (st.n_fast_regions == st.n_regions >= 1) if flags == 0 else (st.n_fast_regions == 0)
    assert len(got) == z.N_PAIRS_GAUSS
    _check_pairs(got, lX, lY, True)
    scaled.close()
    descaled.close()


@pytest.mark.parametrize("flags", [0, sa.FLAG_EXACT])
def test_zymo_whole_read_degenerate_nucleotides_not_ragged(flags):
    # test_DegenerateNucleotides (:920-983): getAlignedPairsUsingAnchors(sM, seq, events, anchors, p, fn, 0, 0)
    r, alpha, k, t10, tab, ax, ay, p = _zymo()
    tp = r["template_params"]
    lX, lY = len(r["ref"]) - (k - 1), r["template_events"].shape[0]
    m = sa.Model.create(alpha, k, t10, _noise_scaled(tab, tp))
    m.set_emission(EM_TWO_DIST)
    jobs = [dict(ref=r["ref"].replace("C", letter), events=r["template_events"], ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"],
                 var=tp["var"], ragged=(0, 0)) for letter in z.N_PAIRS_DEGENERATE]
    b = sa.Batch(m, p, jobs, flags=flags)       # (default ambiguity table: L -> C / E / O, impl/pairwiseAligner.c:32-65)
    b.run()
    assert b.stats().n_fast_regions == 0        # several paths per cell somewhere in the batch: the reference-ordered kernels
    for j, (letter, want) in enumerate(z.N_PAIRS_DEGENERATE.items()):
        got = b.pairs(j)
        assert len(got) == want, (letter, len(got))
        _check_pairs(got, lX, lY, letter != "L")
    b.close()
    m.close()


def test_ragged_ends_change_the_result_and_bad_bits_are_refused():
    # the two booleans are not decoration: a repeat reference and five copies of one event leave the ends of the alignment open,
    # and each flag moves posterior mass there (ragged start: [-inf, 0, 0] instead of [0, -inf, -inf]; ragged end: the gap
    # extension transitions instead of the transitions into the match state, impl/stateMachine.c:1134-1173)
    alpha, k, t10, tab = synth.parse_model_table(os.path.join(GOLDEN, "models", "testModelR9_5mer_acgt_template.model"))
    m = sa.Model.create(alpha, k, t10, tab)
    m.set_emission(EM_TWO_DIST_SCALED_MODEL)
    p = sa.default_params(threshold=0.0001, expansion=2, trace_back=40)
    ev = np.array(SY5, dtype=np.float64).reshape(7, 4)[[5, 5, 5, 5, 5]]
    res = {}
    for rg in ((0, 0), (1, 0), (0, 1), (1, 1)):
        got, _ = _align(m, p, dict(ref="ATATATATATATAT", events=ev, ax=[], ay=[], ragged=rg), 0)
        res[rg] = {(int(q["x"]), int(q["y"])): int(q["prob_e7"]) for q in got}
    assert res[(0, 0)] != res[(1, 0)] and res[(0, 0)] != res[(0, 1)] and res[(1, 0)] != res[(1, 1)] and res[(0, 1)] != res[(1, 1)]
    from signalalign_amd import _capi
    arr, keep = _capi._make_jobs([dict(ref="ACGATATGGACAT", events=ev, ax=[], ay=[])])
    arr[0].ends = 4
    h = _capi.C.c_void_p()
    rc = sa.lib().sa_batch_create(_capi.C.byref(h), m._h, _capi.C.byref(p), arr, 1, sa.default_ambig(), 0, 0)
    assert rc == -1     # SA_EINVAL
    m.close()


@pytest.mark.parametrize("emission", [EM_TWO_DIST, EM_TWO_DIST_SCALED_MODEL])
def test_two_distribution_emission_on_the_register_kernels_against_the_other_kernels_and_the_oracle(emission):
    """Round 6: the two-distribution emissions on the register kernels against the reference-ordered kernels (SA_FLAG_EXACT) and
    the CPU restatement on synthetic reads (dense anchors: the register loop; every 29th anchor: the in-kernel memory-resident
    path), a read of 30 events, both ragged-end settings.  1e-5 on a posterior; rows on one side only within that of the threshold."""
    from oracle import sa_oracle_py as oracle
    import sa_cases as cases
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    m = sa.Model.create(alpha, k, t10, tab)
    m.set_emission(emission)
    om = oracle.Model(alpha, k, t10, tab, emission={EM_TWO_DIST: oracle.EM_TWODIST_DESCALED, EM_TWO_DIST_SCALED_MODEL: oracle.EM_TWODIST}[emission])
    p = sa.default_params(threshold=0.02)
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 1400, 4100) + cases.synthetic_jobs(cases.MODEL_6MER, 1, 30, 4200)
    sparse = dict(jobs[1])
    keep = np.zeros(len(sparse["ax"]), dtype=bool)
    keep[::29] = True
    sparse["ax"], sparse["ay"] = sparse["ax"][keep], sparse["ay"][keep]
    jobs.append(sparse)
    jobs[2] = dict(jobs[2], ragged=(0, 0))
    if emission == EM_TWO_DIST_SCALED_MODEL:    # the model is not scaled to these reads: give them the identity scaling instead
        jobs = [dict(j, scale=1.0, shift=0.0, var=1.0) for j in jobs]
    fast = sa.Batch(m, p, jobs)
    fast.run()
    st = fast.stats()
    assert st.n_fast_regions == st.n_regions == len(jobs)
    exact = sa.Batch(m, p, jobs, flags=sa.FLAG_EXACT)
    exact.run()
    assert exact.stats().n_fast_regions == 0
    worst = 0
    for j, job in enumerate(jobs):
        a, b = fast.pairs(j), exact.pairs(j)
        w, lonely = cases.compare_pairs(a, b, 100, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2 and cases.same_order(a, b), j
        om.set_read_params(job["scale"], job["shift"], job["var"])
        exp = oracle.align(om, job["ref"], job["events"], job["ax"], job["ay"], op, ragged=job.get("ragged", (1, 1)))
        assert np.array_equal(b["prob_e7"], exp["prob_e7"]) and np.array_equal(b["x"], exp["x"])   # EXACT == oracle, bit for bit
        assert len(a) > 0.5 * len(job["events"])
    assert worst <= 10
    fast.close(); exact.close(); m.close()

