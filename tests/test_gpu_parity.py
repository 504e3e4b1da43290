"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star: posteriors within 1e-5 absolute):
  * SA_FLAG_EXACT kernels (reference-ordered un-contracted fp64, host exp): prob_e7 bit-identical,
    same rows, same order.
  * default fast kernels (fma, folded constants, device exp): |dp| <= 1e-5 (100 units of 1e-7); rows may
    differ only where p is within that tolerance of the threshold.  Measured differences are ~1e-9.
  * HDP emissions (device log): 1e-5.
"""
import os

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu

TOL_E7 = 100  # 1e-5 absolute on a posterior


def _models(oracle, path, nhdp=None):
    alpha, k, t10, tab = synth.parse_model_table(path)
    pm = sa.Model.load(path, nhdp)
    om = oracle.Model(alpha, k, t10, tab)
    if nhdp:
        om.load_hdp(nhdp)
    return pm, om


def _run(pm, params, jobs, flags=0, ambig=None):
    b = sa.Batch(pm, params, jobs, ambig=ambig, flags=flags)
    b.run()
    out = [b.pairs(j) for j in range(len(jobs))]
    st = b.stats()
    b.close()
    return out, st


def test_exact_kernels_bit_identical_gaussian(oracle):
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 6, 700) + cases.synthetic_jobs(cases.MODEL_6MER, 2, 2600, 100)
    got, st = _run(pm, p, jobs, flags=sa.FLAG_EXACT)
    assert st.n_fast_regions == 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        assert len(got[j]) == len(exp)
        for f in ("x", "y", "path", "kmer_id", "prob_e7"):
            assert np.array_equal(got[j][f], exp[f]), (j, f)


def test_fast_kernels_within_tolerance_and_same_order(oracle):
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 8, 1500) + cases.synthetic_jobs(cases.MODEL_6MER, 4, 5000, 50)
    got, st = _run(pm, p, jobs)
    assert st.n_fast_regions == st.n_regions
    worst = 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2
        assert cases.same_order(got[j], exp)
        # k_gather's pid[poff[x+1] + path] lookup: the k-mer id of every row both sides hold, in order
        ek = {(int(r["x"]), int(r["y"]), int(r["path"])): int(r["kmer_id"]) for r in exp}
        common = [(int(r["x"]), int(r["y"]), int(r["path"])) in ek for r in got[j]]
        assert sum(common) >= len(exp) - 2
        assert [int(r["kmer_id"]) for r, c in zip(got[j], common) if c] == \
               [ek[(int(r["x"]), int(r["y"]), int(r["path"]))] for r, c in zip(got[j], common) if c]
    print("worst |d prob_e7| fast vs oracle:", worst)
    assert worst <= 10  # measured headroom: differences are ~1e-9, i.e. at most a unit or two of 1e-7


def test_generic_default_matches_fast(oracle):
    # the memory-resident kernels in their RELAX flavour (register-kernel arithmetic, LDS ring, device-side finalisation)
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 1200, 300)
    got, st = _run(pm, p, jobs, flags=sa.FLAG_FORCE_GENERIC)
    assert st.n_fast_regions == 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        w, lonely = cases.compare_pairs(got[j], exp, 10, p.threshold)
        assert cases.same_order(got[j], exp)


def test_bundled_reads_self_reference(oracle):
    # BASELINE config 1 substitute: the three bundled .npRead files, template read as reference, one M run
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    for name, model in [("r9p4_oneD.npRead", cases.MODEL_6MER), ("c2925_ecoli_ch34_read1023.npRead", cases.MODEL_5MER)]:
        pm, om = _models(oracle, model)
        job = cases.npread_job(oracle, name, model)
        exp = cases.oracle_pairs(oracle, om, job, op)
        got, st = _run(pm, p, [job], flags=sa.FLAG_EXACT)
        assert np.array_equal(got[0]["prob_e7"], exp["prob_e7"]) and np.array_equal(got[0]["x"], exp["x"])
        got, st = _run(pm, p, [job])
        w, lonely = cases.compare_pairs(got[0], exp, TOL_E7, p.threshold)
        assert cases.same_order(got[0], exp)


def test_ambiguous_positions_cpg(oracle):
    # config 3 shape: ACEGT model, every CpG cytosine replaced by X with X -> C/E
    pm, om = _models(oracle, cases.MODEL_CPG)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    amb_p = sa.default_ambig({"X": "CE"})
    amb_o = oracle.ambig_map({"X": "CE"})
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, 3, 900, 20, cpg_ambiguous=True)
    assert any("X" in j["ref"] for j in jobs)
    got, st = _run(pm, p, jobs, flags=sa.FLAG_EXACT, ambig=amb_p)
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
        assert len(got[j]) == len(exp)
        for f in ("x", "y", "path", "kmer_id", "prob_e7"):
            assert np.array_equal(got[j][f], exp[f]), (j, f)
        assert exp["path"].max() >= 1
    got, st = _run(pm, p, jobs, ambig=amb_p)
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
        cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        assert cases.same_order(got[j], exp)


def test_hdp_emissions(oracle):
    pm, om = _models(oracle, cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.1)
    op = cases.oracle_params(oracle, p)
    # events drawn from the Gaussian table of the same model, over the ACGT subset of ACEGOT
    jobs = cases.synthetic_jobs(cases.MODEL_R73, 2, 600, 40)
    got, st = _run(pm, p, jobs)
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        assert len(exp) > 100
        cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)


@pytest.mark.parametrize("threshold", [0.1, 0.01])
def test_hdp_emission_plane_reads_of_the_hdp_workload(oracle, threshold):
    """The HDP workload's reads (events drawn from the model's own densities): the emission plane (k_emit_hdp) feeds the
    register sections and, for the read whose anchors are thinned to a sixth, the in-kernel memory-resident stretches (bands
    wider than a wave); a read shorter than one tile of the emission kernel; several forward-storage passes."""
    pm, om = _models(oracle, cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    om.set_to_hdp_expected_values()
    p = sa.default_params(threshold=threshold)
    op = cases.oracle_params(oracle, p)
    jobs = cases.hdp_jobs(3, 1300, 7, table5=pm.table5()) + cases.hdp_jobs(1, 40, 99, table5=pm.table5())
    sparse = dict(jobs[1])
    keep = np.zeros(len(sparse["ax"]), dtype=bool)
    keep[::41] = True
    sparse["ax"], sparse["ay"] = sparse["ax"][keep], sparse["ay"][keep]
    jobs.append(sparse)
    # SA_RING_WIDE=0 keeps the thinned read on the register kernels (their in-kernel memory-resident stretches read the plane
    # too); by default it takes the strip kernels (round 4: tests/test_gpu_hdp_ambig.py), compared below as well
    os.environ["SA_RING_WIDE"] = "0"
    try:
        got, st = _run(pm, p, jobs)
    finally:
        del os.environ["SA_RING_WIDE"]
    assert st.n_fast_regions == st.n_regions
    exp = [cases.oracle_pairs(oracle, om, job, op) for job in jobs]
    for j in range(len(jobs)):
        assert len(exp[j]) > (0.02 if threshold >= 0.1 else 3.0) * len(jobs[j]["events"])
        cases.compare_pairs(got[j], exp[j], TOL_E7, p.threshold)
    got_d, st_d = _run(pm, p, jobs)
    assert st_d.n_fast_regions + st_d.n_ring_regions == st_d.n_regions and st_d.n_strip_regions >= 1
    for j in range(len(jobs)):
        cases.compare_pairs(got_d[j], exp[j], TOL_E7, p.threshold)
    os.environ["SA_F_BUDGET_CELLPATHS"] = "150000"
    try:
        got2, st2 = _run(pm, p, jobs)
    finally:
        del os.environ["SA_F_BUDGET_CELLPATHS"]
    assert st2.n_chunks >= 2
    for j in range(len(jobs)):
        assert np.array_equal(got2[j], got_d[j])


def test_edge_cases(oracle):
    pm, om = _models(oracle, cases.MODEL_5MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_5MER)
    base = synth.make_read(900, 400, alpha, k, tab)
    jobs = [
        dict(base, ax=np.zeros(0, dtype=np.int64), ay=np.zeros(0, dtype=np.int64)),           # no anchors at all
        dict(ref=base["ref"][:k + 2], events=base["events"][:4], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0),  # tiny
        dict(ref=base["ref"][:60], events=base["events"][:0], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0),    # no events
        dict(ref=base["ref"][:k - 1], events=base["events"][:9], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0),  # no k-mers
        dict(ref="", events=base["events"][:0], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0),                   # empty
        base,
    ]
    for flags in (sa.FLAG_EXACT, 0):
        got, st = _run(pm, p, jobs, flags=flags)
        for j, job in enumerate(jobs):
            exp = cases.oracle_pairs(oracle, om, job, op)
            if flags:
                assert np.array_equal(got[j]["prob_e7"], exp["prob_e7"]), j
                assert np.array_equal(got[j]["x"], exp["x"]) and np.array_equal(got[j]["y"], exp["y"]), j
            else:
                cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)


def test_threshold_zero_and_one(oracle):
    pm, om = _models(oracle, cases.MODEL_5MER)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_5MER)
    job = synth.make_read(901, 150, alpha, k, tab)
    for thr in (0.0, 1.0):
        p = sa.default_params(threshold=thr)
        op = cases.oracle_params(oracle, p)
        exp = cases.oracle_pairs(oracle, om, job, op)
        got, st = _run(pm, p, [job], flags=sa.FLAG_EXACT)   # threshold 0 overflows the first candidate plan: retried
        assert len(got[0]) == len(exp)
        assert np.array_equal(got[0]["prob_e7"], exp["prob_e7"])


def test_deferred_creation_same_bytes(oracle):
    """sa_batch_create_deferred: the first half of the creation on the caller's thread, the second on the batch's first use
    (run, the thread of start, stats).  Same pairs, byte for byte, as sa_batch_create; two deferred batches alive at once with
    the second one created while the first runs (the bench's loop for workloads that fill the device); a batch that is created
    and never used goes away cleanly; a batch the planning kernels do not take (SA_FLAG_EXACT) is created completely at once;
    a read no planner takes (an anchor outside the matrix) is an error in both modes."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 5, 900, 300) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 3, 2500, 600)
    more = cases.synthetic_jobs(cases.MODEL_6MER, 4, 1100, 311)
    got, st = _run(pm, p, jobs)
    got2, _ = _run(pm, p, more)
    a = sa.Batch(pm, p, jobs, deferred=True, flags=sa.FLAG_DEVICE_TO_ITSELF)
    a.start()
    b = sa.Batch(pm, p, more, deferred=True, flags=sa.FLAG_DEVICE_TO_ITSELF)     # first half while `a` runs
    b.prepare()                                       # sa_batch_prepare: plan and launch lists too, before `a` has finished
    b.prepare()                                       # (once; the second call returns the same code)
    a.wait()
    sta = a.stats()
    assert (sta.cells_forward, sta.cells_backward, sta.n_regions, sta.n_strip_regions) == \
           (st.cells_forward, st.cells_backward, st.n_regions, st.n_strip_regions)
    for j in range(len(jobs)):
        assert np.array_equal(a.pairs(j), got[j]), j
    a.close()
    assert b.stats().n_regions == len(more)          # statistics before the run: the second half runs here
    b.run()
    for j in range(len(more)):
        assert np.array_equal(b.pairs(j), got2[j]), j
    b.close()
    sa.Batch(pm, p, jobs, deferred=True).close()      # created, never used
    c = sa.Batch(pm, p, jobs, deferred=True)
    c.prepare()
    c.close()                                         # prepared, never used
    e = sa.Batch(pm, p, more, deferred=True, flags=sa.FLAG_EXACT)   # not a batch for the planning kernels: complete at once
    e.run()
    exact, _ = _run(pm, p, more, flags=sa.FLAG_EXACT)
    for j in range(len(more)):
        assert np.array_equal(e.pairs(j), exact[j]), j
    e.close()
    bad = dict(more[1])
    bad["ax"] = np.array(list(bad["ax"][:-1]) + [len(bad["ref"]) + 5], dtype=np.int64)   # last anchor beyond the reference
    for deferred in (False, True):
        try:
            c = sa.Batch(pm, p, [more[0], bad], deferred=deferred)
            if deferred:
                try:
                    c.prepare()                       # the error arrives here already ...
                    raise AssertionError("sa_batch_prepare accepted a read no planner takes")
                except sa.SaError:
                    pass
            c.run()                                   # ... and again on the first use
            raised = None
        except sa.SaError as err:
            raised = err.code
        assert raised is not None and raised != 0, deferred


def test_threshold_zero_default_flags(oracle):
    """Threshold 0 keeps every band cell: with default flags the batch is routed to the reference-ordered kernels (the
    default kernels' candidate filter has no lower bound at log 0) and must list exactly the oracle's rows -- dense anchors
    (register kernels otherwise) and sparse ones (strip kernels otherwise)."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params(threshold=0.0)
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 2, 500, 300) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 1, 900, 600)
    got, st = _run(pm, p, jobs)
    assert st.n_fast_regions == 0 and st.n_strip_regions == 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        assert len(got[j]) == len(exp), j
        for f in ("x", "y", "prob_e7"):
            assert np.array_equal(got[j][f], exp[f]), (j, f)


def test_sparse_anchors_wide_bands(oracle, monkeypatch):
    # realistic guide alignments: anchor-free windows widen the band beyond 64 cells, the register kernels
    # must hand over to the memory-resident path and back
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 3000, 700, thin_anchors=0.35)
    exp = [cases.oracle_pairs(oracle, om, job, op) for job in jobs]
    for ring_wide in ("0", "1"):     # 0: register kernels with their memory-resident hand-over; 1: default routing
        monkeypatch.setenv("SA_RING_WIDE", ring_wide)
        got, st = _run(pm, p, jobs)
        assert st.n_fast_regions + st.n_ring_regions == st.n_regions
        if ring_wide == "0":
            assert st.n_fast_regions == st.n_regions
        for j, job in enumerate(jobs):
            w, lonely = cases.compare_pairs(got[j], exp[j], TOL_E7, p.threshold)
            assert cases.same_order(got[j], exp[j])


def test_ring_kernels_wide_bands_every_read_against_the_oracle(oracle, monkeypatch):
    """Anchors as sparse as a real guide alignment leaves them (tests/golden/cigars/ecoli_minus_strand.cigar, -m 14):
    bands of 100-300 cells.  Such regions run on the LDS-ring kernels (one lane per cell, rows of 128 / 256 / 512
    cell-paths by the widest diagonal of the region); every read is compared with the CPU restatement, and with the
    register kernels' own wide-band path (SA_RING_WIDE=0), which computes the same arithmetic in another order."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.realistic_anchor_jobs(cases.MODEL_6MER, 10, 2500, 600)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    for n_ev, idx in ((260, 41), (520, 42), (90, 43)):   # no anchors at all: the band is the whole matrix (rows up to ~400 cells)
        r = synth.make_read(idx, n_ev, alpha, k, tab)
        jobs.append(dict(r, ax=np.zeros(0, dtype=np.int64), ay=np.zeros(0, dtype=np.int64)))
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 300)      # dense anchors: these stay on the register kernels
    got, st = _run(pm, p, jobs)
    assert st.n_ring_regions >= 12 and st.n_fast_regions >= 3 and st.n_ring_regions + st.n_fast_regions == st.n_regions
    worst = 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2 and cases.same_order(got[j], exp), j
    assert worst <= 10
    for waves in ("1", "2", "4"):                     # workgroup size of the forward ring kernel: same bytes
        monkeypatch.setenv("SA_RING_WAVES", waves)
        again, _ = _run(pm, p, jobs)
        for j in range(len(jobs)):
            assert np.array_equal(again[j], got[j]), (waves, j)
    monkeypatch.delenv("SA_RING_WAVES")
    monkeypatch.setenv("SA_RING_WIDE", "0")
    fast, st2 = _run(pm, p, jobs)
    assert st2.n_ring_regions == 0
    for j in range(len(jobs)):
        cases.compare_pairs(got[j], fast[j], 10, p.threshold)


def test_strip_kernels_wide_bands_bit_identical_to_the_ring_kernels(oracle, monkeypatch):
    """One-path regions with wide bands run on the strip kernels (sa_strip.inc: a lane is a reference column, strips of 64
    columns, seams through HBM, candidates by a second pass over forward + backward).  Per cell they do the ring kernels'
    arithmetic in the ring kernels' order, so every pair -- rows, order, prob_e7 -- must equal the ring kernels' (SA_STRIP=0),
    and both are within the tolerance of the CPU restatement.  Cases: anchors of a real guide alignment, no anchors at all
    (bands of several hundred cells, one strip holds whole diagonals), reads shorter than one strip, several traceback
    segments per read, ragged ends off (the default)."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = cases.realistic_anchor_jobs(cases.MODEL_6MER, 8, 2500, 600) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 2, 5200, 700)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    for n_ev, idx in ((260, 41), (520, 42), (90, 43), (35, 44)):
        r = synth.make_read(idx, n_ev, alpha, k, tab)
        jobs.append(dict(r, ax=np.zeros(0, dtype=np.int64), ay=np.zeros(0, dtype=np.int64)))
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 2, 900, 300)      # dense anchors: register kernels
    got, st = _run(pm, p, jobs)
    assert st.n_ring_regions >= 12 and st.n_strip_regions == st.n_ring_regions
    monkeypatch.setenv("SA_STRIP", "0")
    ring, st2 = _run(pm, p, jobs)
    for waves in ("1", "2", "4"):          # the ring kernels' workgroup shapes (1-8 cells per thread and diagonal): same bytes
        monkeypatch.setenv("SA_RING_WAVES", waves)
        ring_w, _ = _run(pm, p, jobs)
        for j in range(len(jobs)):
            assert np.array_equal(ring_w[j], ring[j]), (waves, j)
    monkeypatch.delenv("SA_RING_WAVES")
    monkeypatch.delenv("SA_STRIP")
    assert st2.n_strip_regions == 0 and st2.n_ring_regions == st.n_ring_regions
    for j in range(len(jobs)):
        assert len(got[j]) == len(ring[j]), (j, len(got[j]), len(ring[j]))
        for f in ("x", "y", "path", "kmer_id", "prob_e7"):
            assert np.array_equal(got[j][f], ring[j][f]), (j, f)
    worst = 0
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
        worst = max(worst, w)
        assert lonely <= 2 and cases.same_order(got[j], exp), j
    assert worst <= 10
    again, _ = _run(pm, p, jobs)                                      # same bytes on a second batch (seams, atomics, planes)
    for j in range(len(jobs)):
        assert np.array_equal(again[j], got[j]), j
    # round 4: `got` came from the one-pass backward sweep (candidates against the traceback's speculative total, survivors put in
    # order by k_gather_strip); the two-pass sweep of round 2 (SA_STRIP_PASSES=2) gives the same bytes -- at several thresholds
    # (a low one multiplies the candidates per diagonal, threshold 0.5 leaves diagonals without any) and with short tracebacks
    for kw in (dict(), dict(threshold=0.0005), dict(threshold=0.5), dict(expansion=20, trace_back=30, min_diags=150)):
        q = sa.default_params(**kw)
        one, st1 = _run(pm, q, jobs)
        monkeypatch.setenv("SA_STRIP_PASSES", "2")
        two, st2p = _run(pm, q, jobs)
        monkeypatch.delenv("SA_STRIP_PASSES")
        assert st1.n_strip_regions == st2p.n_strip_regions and st1.n_strip_regions >= 12
        for j in range(len(jobs)):
            assert np.array_equal(one[j], two[j]), (kw, j, len(one[j]), len(two[j]))


def test_ring_kernels_ambiguous_positions_every_read_against_the_oracle(oracle, monkeypatch):
    """BASELINE configs[2] shape (ACEGT model, every CpG cytosine X -> C/E: 1, 2, 4 or 8 paths per cell) and the R7.3
    ACEGOT model with the default table's three-way code L -> C/E/O: the ring kernels with per-path neighbour records
    against the CPU restatement for every read, and against the memory-resident kernels (SA_RING=0)."""
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    for model, amb_tab, jobs in (
            (cases.MODEL_CPG, {"X": "CE"}, cases.synthetic_jobs(cases.MODEL_CPG, 6, 1400, 20, cpg_ambiguous=True) +
             cases.realistic_anchor_jobs(cases.MODEL_CPG, 2, 1200, 77)),
            # sparse variant positions (every 9th / 30th CpG cytosine): regions below 1.3 paths per column take the ring kernels'
            # other instance, which tests per wave whether a second predecessor / successor exists at all (round 4); one batch
            # holds both kinds
            (cases.MODEL_CPG, {"X": "CE"}, cases.synthetic_jobs(cases.MODEL_CPG, 3, 1400, 320, cpg_ambiguous=True, cpg_every=9) +
             cases.synthetic_jobs(cases.MODEL_CPG, 2, 1100, 330, cpg_ambiguous=True, cpg_every=30) +
             cases.synthetic_jobs(cases.MODEL_CPG, 1, 900, 340, cpg_ambiguous=True) + [None, None]),
            (cases.MODEL_R73, None, None)):
        pm, om = _models(oracle, model)
        amb_p = sa.default_ambig(amb_tab)
        amb_o = oracle.ambig_map(amb_tab)
        if jobs is None:
            jobs = []
            for j, job in enumerate(cases.synthetic_jobs(model, 3, 600, 50)):
                ref = list(job["ref"])
                for i in range(7 + j, len(ref) - 6, 23):
                    if ref[i] == "C":
                        ref[i] = "L"
                jobs.append(dict(job, ref="".join(ref)))
        elif jobs[-1] is None:
            jobs = jobs[:-2]
        else:
            jobs[-1] = dict(jobs[-1], ref=jobs[-1]["ref"].replace("CG", "XG"))
            jobs[-2] = dict(jobs[-2], ref=jobs[-2]["ref"].replace("CG", "XG"))
        got, st = _run(pm, p, jobs, ambig=amb_p)
        assert st.n_ring_regions == st.n_regions == len(jobs)
        worst = 0
        for j, job in enumerate(jobs):
            exp = cases.oracle_pairs(oracle, om, job, op, ambig=amb_o)
            assert exp["path"].max() >= 1
            w, lonely = cases.compare_pairs(got[j], exp, TOL_E7, p.threshold)
            worst = max(worst, w)
            assert lonely <= 2 and cases.same_order(got[j], exp), j
            ek = {(int(r["x"]), int(r["y"]), int(r["path"])): int(r["kmer_id"]) for r in exp}
            assert all(ek.get((int(r["x"]), int(r["y"]), int(r["path"])), int(r["kmer_id"])) == int(r["kmer_id"]) for r in got[j])
        assert worst <= 10
        # every workgroup shape of the ring kernels (1, 2, 4 waves: one, two, four or eight cell-paths per thread and
        # diagonal -- the candidate prefix across waves has a vector path for one chunk per wave and a packed-word path for
        # several) gives the same bytes
        for waves in ("1", "2", "4"):
            monkeypatch.setenv("SA_RING_WAVES", waves)
            again, _ = _run(pm, p, jobs, ambig=amb_p)
            for j in range(len(jobs)):
                assert np.array_equal(again[j], got[j]), (waves, j)
        monkeypatch.delenv("SA_RING_WAVES")
        monkeypatch.setenv("SA_RING", "0")
        old, st2 = _run(pm, p, jobs, ambig=amb_p)
        monkeypatch.delenv("SA_RING")
        assert st2.n_ring_regions == 0
        for j in range(len(jobs)):
            cases.compare_pairs(got[j], old[j], 10, p.threshold)


def test_split_regions_and_chunked_forward_storage(oracle, monkeypatch):
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    big = synth.make_read(7, 14000, alpha, k, tab)
    hole = (big["ax"] > 1500) & (big["ax"] < 6500)
    big["ax"], big["ay"] = big["ax"][~hole], big["ay"][~hole]
    jobs = [big] + cases.synthetic_jobs(cases.MODEL_6MER, 5, 1000, 800)
    exp = [cases.oracle_pairs(oracle, om, job, op) for job in jobs]
    got, st = _run(pm, p, jobs)
    assert st.n_regions == len(jobs) + 1 and st.n_chunks == 1
    for j in range(len(jobs)):
        cases.compare_pairs(got[j], exp[j], TOL_E7, p.threshold)
        assert cases.same_order(got[j], exp[j])
    # force several passes over the forward storage
    monkeypatch.setenv("SA_F_BUDGET_CELLPATHS", "200000")
    got2, st2 = _run(pm, p, jobs)
    assert st2.n_chunks > 1
    for j in range(len(jobs)):
        assert np.array_equal(got2[j], got[j])
    # several result groups per pass (pipelined result copy), alone and combined with several passes; a second
    # run() of the same batch takes the overlapped-copy path (the pinned buffer is sized by the first)
    for env in ({"SA_GROUPS": "3"}, {"SA_GROUPS": "2", "SA_F_BUDGET_CELLPATHS": "200000"}):
        monkeypatch.delenv("SA_F_BUDGET_CELLPATHS", raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        b = sa.Batch(pm, p, jobs + [dict(ref="ACGTACGT", events=np.zeros(0), ax=[], ay=[])])
        for _ in range(2):
            b.run()
            for j in range(len(jobs)):
                assert np.array_equal(b.pairs(j), got[j]), (env, j)
            assert b.n_pairs(len(jobs)) == 0
        b.close()


def test_round_trip_properties_full_size(oracle):
    # size-independent properties on a BASELINE-sized read (10k events): per event the posteriors over
    # reference positions sum to <= 1 (+ logAdd approximation slack), rows are ordered, coordinates in range
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 4, 10000, 5000)
    got, st = _run(pm, p, jobs)
    for j, job in enumerate(jobs):
        g = got[j]
        lX, lY = len(job["ref"]) - 5, len(job["events"])
        assert g["x"].min() >= 0 and g["x"].max() < lX and g["y"].min() >= 0 and g["y"].max() < lY
        s = g["x"].astype(np.int64) + g["y"]
        assert np.all(np.diff(s) >= 0)
        per_event = np.bincount(g["y"], weights=g["prob_e7"] / 1e7, minlength=lY)
        assert per_event.max() <= 1.0 + 5e-3
        assert per_event.mean() > 0.4           # extra events (gapY) carry no match posterior; the rest do
        assert np.all(g["prob_e7"] >= int(p.threshold * 1e7)) and np.all(g["prob_e7"] <= 10000000)
    # idempotence: a second run of the same batch gives identical bytes
    got2, _ = _run(pm, p, jobs)
    for a, b in zip(got, got2):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name,npread,model,nhdp", [("r9_5mer", "c2925_ecoli_ch34_read1023.npRead", cases.MODEL_5MER, None),
                                                     ("r94_6mer", "r9p4_oneD.npRead", cases.MODEL_6MER, None),
                                                     ("r73_acegot", "ZymoC_ch_1_file1.npRead", cases.MODEL_R73, None),
                                                     ("r73_acegot_hdp", "ZymoC_ch_1_file1.npRead", cases.MODEL_R73, cases.NHDP)])
def test_against_committed_expected_outputs(name, npread, model, nhdp):
    # no oracle in this test: inputs from the reference's fixture files, expected pairs from tests/golden/expected
    want = np.load(os.path.join(cases.GOLDEN, "expected", name + ".npz"))
    pm = sa.Model.load(model, nhdp)
    if nhdp:
        pm.set_to_hdp_expected_values()
    # the job as make_expected.py built it, from the stored anchors and parameters
    lines = open(os.path.join(cases.GOLDEN, "npReads", npread)).read().split("\n")
    read = lines[2].strip()
    ev = np.array(lines[7].split(), dtype=np.float64).reshape(-1, 4)
    emap = np.array(lines[3].split(), dtype=np.int64)
    L = len(read)
    lo, hi = int(emap[0]), int(emap[L - 1])
    # drift correction of the events is part of the parameter estimation: redo it through the library
    plain = sa.Model.load(model)                      # estimation starts from the table as loaded (no HDP means yet)
    est = sa.estimate_params(plain, np.array(plain.table5(), dtype=np.float64, copy=True), emap, ev, read)
    assert abs(est["scale"] - float(want["scale"])) < 1e-12 and abs(est["var"] - float(want["var"])) < 1e-12
    job = dict(ref=read, events=np.ascontiguousarray(ev[lo:hi]), ax=want["ax"], ay=want["ay"], scale=float(want["scale"]),
               shift=float(want["shift"]), var=float(want["var"]))
    p = sa.default_params(threshold=float(want["threshold"]))
    exp = np.zeros(len(want["x"]), dtype=sa.PAIR_DTYPE)
    for f in ("x", "y", "path", "kmer_id", "prob_e7"):
        exp[f] = want[f]
    if not nhdp:
        got, _ = _run(pm, p, [job], flags=sa.FLAG_EXACT)
        assert np.array_equal(got[0], exp)               # bit-identical rows, values and order
    got, st = _run(pm, p, [job])
    assert st.n_fast_regions == st.n_regions
    cases.compare_pairs(got[0], exp, TOL_E7, p.threshold)
    assert cases.same_order(got[0], exp)


@pytest.mark.parametrize("name", ["zymo_lastz_C", "zymo_lastz_E", "zymo_lastz_O", "zymo_lastz_L", "zymo_lastz_hdp"])
def test_reference_whole_read_jobs(name):
    """The reference's whole-read test jobs (tests/stateMachineTests.c:842-983: ZymoC x ZymoRef, anchors from the
    reference's lastz subprocess committed under tests/golden/cigars, banding defaults expansion 20 / traceBack 40)
    through the HIP path, against committed outputs (no oracle in this test).  zymo_lastz_hdp is
    test_sm3Hdp_getAlignedPairsWithBanding itself: the reference expects exactly 1217 pairs at threshold 0.1.
    zymo_lastz_L has three paths per C (C/E/O): the ambiguity expansion on a real read."""
    import zymo_wholeread as z
    want = np.load(os.path.join(cases.GOLDEN, "expected", name + ".npz"))
    hdp = name.endswith("_hdp")
    r = z.read_fixture()
    ax, ay = z.remapped_anchors()
    tp = r["template_params"]
    ref = r["ref"] if hdp else r["ref"].replace("C", name[-1])
    ev = z.hdp_test_events(r) if hdp else r["template_events"]
    job = dict(ref=ref, events=np.ascontiguousarray(ev), ax=ax, ay=ay, scale=tp["scale"], shift=tp["shift"], var=tp["var"])
    b = z.BANDING
    p = sa.default_params(threshold=float(want["threshold"]), expansion=b["expansion"], trace_back=b["trace_back"],
                          min_diags=b["min_diags"], split=b["split"])
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP if hdp else None)
    exp = np.zeros(len(want["x"]), dtype=sa.PAIR_DTYPE)
    for f in ("x", "y", "path", "kmer_id", "prob_e7"):
        exp[f] = want[f]
    if hdp:
        assert len(exp) == z.N_PAIRS_HDP_AT_0p1
    got, _ = _run(pm, p, [job], flags=sa.FLAG_EXACT)
    if hdp:   # device log: 1e-5
        w, lonely = cases.compare_pairs(got[0], exp, TOL_E7, p.threshold)
        assert len(got[0]) == z.N_PAIRS_HDP_AT_0p1, (len(got[0]), lonely)
    else:
        assert np.array_equal(got[0], exp)
    got, _ = _run(pm, p, [job])
    cases.compare_pairs(got[0], exp, TOL_E7, p.threshold)
    assert cases.same_order(got[0], exp)
    if hdp:
        assert len(got[0]) == z.N_PAIRS_HDP_AT_0p1


def test_working_storage_that_does_not_fit_is_planned_again(monkeypatch):
    """A batch whose working storage does not fit the device (a deferred batch whose budget dates from before another batch took
    the memory; candidate slots of an HDP model at a low threshold) re-packs its forward storage into more passes and tries
    again instead of returning SA_ENOMEM.  The failure is injected (SA_TEST_FAIL_WORKING_ALLOC: attempts fail while the plan has
    fewer passes); a second batch is alive meanwhile.  Same bytes as the unconstrained run, for device- and host-planned batches
    and for a batch created in two halves."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 12, 1500, 900) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 3, 2200, 950)
    ref = sa.Batch(pm, p, jobs)
    ref.run()
    assert ref.stats().n_chunks == 1
    want = [ref.pairs(j) for j in range(len(jobs))]
    monkeypatch.setenv("SA_TEST_FAIL_WORKING_ALLOC", "3")
    for kw in (dict(), dict(deferred=True, flags=sa.FLAG_DEVICE_TO_ITSELF)):
        b = sa.Batch(pm, p, jobs, **kw)        # (`ref` is still alive and holds its storage)
        b.run()
        assert b.stats().n_chunks >= 3
        for j in range(len(jobs)):
            assert np.array_equal(b.pairs(j), want[j]), (kw, j)
        b.close()
    monkeypatch.setenv("SA_DEVICE_PLAN", "0")    # the host planner's batch
    b = sa.Batch(pm, p, jobs)
    b.run()
    assert b.stats().n_chunks >= 3
    for j in range(len(jobs)):
        assert np.array_equal(b.pairs(j), want[j]), j
    b.close()
    ref.close()


def test_storage_reuse_across_batches_and_release(oracle):
    """Batches take their storage from caching allocators: a second, DIFFERENT batch built from the blocks of a destroyed
    one (stale bytes in every buffer) gives the same pairs as in a fresh process state, and sa_pool_release() leaves the
    library usable."""
    import ctypes as C
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs_a = cases.synthetic_jobs(cases.MODEL_6MER, 6, 1100, 50)
    jobs_b = cases.synthetic_jobs(cases.MODEL_6MER, 5, 1300, 900)
    sa.lib().sa_pool_release()
    fresh_b, _ = _run(pm, p, jobs_b)
    sa.lib().sa_pool_release()
    _run(pm, p, jobs_a)                       # its blocks are parked when it is destroyed ...
    reused_b, _ = _run(pm, p, jobs_b)         # ... and serve this one
    for x, y in zip(fresh_b, reused_b):
        assert np.array_equal(x, y)
    sa.lib().sa_pool_release()
    again_b, _ = _run(pm, p, jobs_b)
    for x, y in zip(fresh_b, again_b):
        assert np.array_equal(x, y)


def test_start_wait_overlaps_batches(oracle):
    """sa_batch_start / sa_batch_wait: two batches in flight (the second planned while the first is on the GPU) give what
    two serial runs give; waiting twice or starting twice is a state error."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs_a = cases.synthetic_jobs(cases.MODEL_6MER, 5, 1200, 10)
    jobs_b = cases.synthetic_jobs(cases.MODEL_6MER, 4, 900, 700)
    want_a, _ = _run(pm, p, jobs_a)
    want_b, _ = _run(pm, p, jobs_b)
    a = sa.Batch(pm, p, jobs_a)
    a.start()
    with pytest.raises(sa.SaError):
        a.start()
    b = sa.Batch(pm, p, jobs_b)
    b.start()
    a.wait()
    b.wait()
    with pytest.raises(sa.SaError):
        a.wait()
    for j in range(len(jobs_a)):
        assert np.array_equal(a.pairs(j), want_a[j])
    for j in range(len(jobs_b)):
        assert np.array_equal(b.pairs(j), want_b[j])
    a.close()
    c = sa.Batch(pm, p, jobs_a)
    c.start()
    c.close()                                  # destroying a started batch joins it first
    b.close()


def test_speculative_bound_that_is_too_high_repeats_the_pass(oracle, monkeypatch, capfd):
    """The backward sweeps bound their candidates by the traceback's speculative total minus a slack;
    k_finalize checks every exact total against that bound and the pass is repeated with four times the slack when one falls
    below.  Forced here with a slack (1e-4) the totals' drift exceeds (~1e-2 per traceback): register, strip and ring kernels, the
    same bytes as with the default slack."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 4, 1500, 77) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 2, 2500, 91)
    want, st = _run(pm, p, jobs)
    assert st.n_fast_regions >= 4 and st.n_strip_regions >= 1
    pc = sa.Model.load(cases.MODEL_CPG)
    amb = sa.default_ambig({"X": "CE"})
    cjobs = cases.synthetic_jobs(cases.MODEL_CPG, 2, 1200, 5, cpg_ambiguous=True)
    cwant, cst = _run(pc, p, cjobs, ambig=amb)
    assert cst.n_ring_regions == 2
    capfd.readouterr()
    monkeypatch.setenv("SA_TEST_SPEC_SLACK", "1e-4")
    got, _ = _run(pm, p, jobs)
    cgot, _ = _run(pc, p, cjobs, ambig=amb)
    err = capfd.readouterr().err
    assert err.count("a speculative candidate bound was too high") >= 2, err[-400:]
    for j in range(len(jobs)):
        assert np.array_equal(got[j], want[j]), j
    for j in range(len(cjobs)):
        assert np.array_equal(cgot[j], cwant[j]), j


def test_released_batches_keep_their_results_and_a_nan_event_is_reported(oracle, capfd):
    """sa_batch_release_device (round 5): a finished batch's HBM goes back to the allocator while its packed pairs, offsets and
    statistics stay readable; running it again is SA_ESTATE, the chained MEA step still works (pairs uploaded again).  And a read
    whose event means hold a NaN no longer comes back as an empty alignment without a word: k_spec_match sees forward values that
    are not numbers and sa_batch_run returns SA_EINVAL."""
    pm, om = _models(oracle, cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 400)
    b = sa.Batch(pm, p, jobs)
    with pytest.raises(sa.SaError):
        b.release_device()                      # before it has run
    b.run()
    want = [b.pairs(j).copy() for j in range(3)]
    mea_before = b.mea()
    dev = b.stats().device_bytes
    b.release_device()
    b.release_device()                          # idempotent
    assert b.stats().device_bytes == dev
    for j in range(3):
        assert np.array_equal(b.pairs(j), want[j]) and np.array_equal(b.pairs16(j), sa.Batch.pairs16(b, j))
    with pytest.raises(sa.SaError) as ei:
        b.run()
    assert ei.value.code == -7
    with pytest.raises(sa.SaError) as ei:       # ... and sa_batch_start says so itself (not a thread that fails at sa_batch_wait)
        b.start()
    assert ei.value.code == -7
    mea_after = b.mea()
    for x, y in zip(mea_before, mea_after):
        assert np.array_equal(x[0], y[0]) and x[1] == y[1] and x[2] == y[2]
    b.close()
    bad = dict(jobs[0])
    ev = np.array(bad["events"], dtype=np.float64).copy()
    ev.reshape(len(ev), -1)[len(ev) // 2, 0] = np.nan
    bad["events"] = ev
    capfd.readouterr()
    b2 = sa.Batch(pm, p, [jobs[1], bad])
    with pytest.raises(sa.SaError) as ei:
        b2.run()
    assert ei.value.code == -1 and "not a finite number" in capfd.readouterr().err
    b2.close()


def test_eight_byte_result_records(oracle):
    """SA_FLAG_PAIRS8: the same pairs as the 16-byte records in the same order, (x, y, prob_e7) only -- register, strip and
    memory-resident kernels, device and host finalisation; refused where a cell may hold several paths or a coordinate does not fit
    20 bits; the 16-byte accessors refuse such a batch."""
    pm = sa.Model.load(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 5, 1400, 31) + cases.realistic_anchor_jobs(cases.MODEL_6MER, 3, 2500, 77)
    p = sa.default_params(threshold=0.01)
    for flags in (0, sa.FLAG_EXACT, sa.FLAG_FORCE_GENERIC):
        a = sa.Batch(pm, p, jobs, flags=flags)
        a.run()
        b = sa.Batch(pm, p, jobs, flags=flags | sa.FLAG_PAIRS8)
        b.run()
        view, first = b.results_view()
        assert view.shape == (int(first[-1]), 1)
        for j in range(len(jobs)):
            x, y = a.pairs(j), b.pairs8(j)
            assert len(x) == len(y) == b.n_pairs(j) and len(x) > 100
            assert np.array_equal(x["x"], y["x"]) and np.array_equal(x["y"], y["y"]) and np.array_equal(x["prob_e7"], y["prob_e7"])
            assert a.all_pairs_summary(j) == b.all_pairs_summary(j)
        with pytest.raises(sa.SaError):
            b.pairs(0)
        with pytest.raises(sa.SaError):
            a.pairs8(0)
        with pytest.raises(sa.SaError):
            b.mea()
        a.close(); b.close()
    cm = sa.Model.load(cases.MODEL_CPG)
    cj = cases.synthetic_jobs(cases.MODEL_CPG, 2, 600, 5)
    s = list(cj[0]["ref"]); s[s.index("C", 50)] = "X"; cj[0]["ref"] = "".join(s)
    with pytest.raises(sa.SaError) as ei:
        sa.Batch(cm, p, cj, ambig=sa.default_ambig({"X": "CE"}), flags=sa.FLAG_PAIRS8).run()
    assert ei.value.code == -8   # SA_EUNSUPPORTED


def test_hdp_emission_kernels_without_a_hot_row(oracle):
    """k_emit_hdp / k_emit_hdp_ring stage the row most k-mers resolve to in LDS when a model has one (the bundled model: the base
    process's, 99 % of the k-mers); a model without such a row takes the loops that read every cell's coefficients from memory.
    SA_HDP_HOT=0 forces those on the bundled model: the same pairs, byte for byte, register, strip and ring regions."""
    pm, om = _models(oracle, cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.05)
    jobs = cases.hdp_jobs(3, 900, 11, table5=pm.table5())
    sparse = dict(jobs[1])
    keep = np.zeros(len(sparse["ax"]), dtype=bool)
    keep[::37] = True
    sparse["ax"], sparse["ay"] = sparse["ax"][keep], sparse["ay"][keep]
    amb = dict(jobs[2])
    amb["ref"] = amb["ref"].replace("CG", "XG")
    jobs = jobs + [sparse, amb]
    ambig = sa.default_ambig({"X": "CE"})
    got, st = _run(pm, p, jobs, ambig=ambig)
    assert st.n_fast_regions >= 1 and st.n_ring_regions >= 1
    os.environ["SA_HDP_HOT"] = "0"
    try:
        got2, _ = _run(pm, p, jobs, ambig=ambig)
    finally:
        del os.environ["SA_HDP_HOT"]
    for a, b in zip(got, got2):
        assert len(a) > 50 and np.array_equal(a, b)
