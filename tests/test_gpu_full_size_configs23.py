"""BASELINE configs[2] and configs[3] at the sizes BASELINE.json names, through the C ABI:

  configs[2]  10 000 synthetic 5000-event reads, R9.4 6-mer CpG model (ACEGT), every CpG cytosine ambiguous (C/E): 1-8 paths
              per cell, ring kernels with per-path neighbour records, one batch takes most of the card;
  configs[3]  5 000 synthetic 5000-event reads, HDP emissions (templateSingleLevelFixed.nhdp over the R7.3 ACEGOT model):
              emission plane (k_emit_hdp) + register kernels; threshold 0.1 as the reference's own HDP test uses
              (tests/stateMachineTests.c:912) and the default 0.01.

Too large for the CPU restatement to walk, so the checks are the size-independent ones: posteriors are probabilities in
TSV order; a second run returns the same bytes (checksum of per-read checksums); reads run alone in a small batch give the
same bytes as among thousands; and two reads of each agree with the CPU restatement within the 1e-5 bar.
"""
import zlib

import numpy as np
import pytest

import signalalign_amd as sa

import sa_cases as cases

pytestmark = pytest.mark.gpu
N_EVENTS = 5000


def _digest(b, n):
    per_read = [zlib.crc32(b.pairs(j).tobytes()) for j in range(n)]
    return zlib.crc32(np.asarray(per_read, dtype=np.uint32).tobytes()), per_read


def _check_probabilities(b, jobs, threshold, step, min_pairs_per_event):
    for j in range(0, len(jobs), step):
        pr = b.pairs(j)
        assert len(pr) >= min_pairs_per_event * len(jobs[j]["events"]), j
        if len(pr) == 0:
            continue
        assert pr["prob_e7"].min() >= int(threshold * 1e7) and pr["prob_e7"].max() <= 10_000_000
        assert np.all(np.diff(pr["x"] + pr["y"]) >= 0)
        assert pr["x"].min() >= 0 and pr["y"].min() >= 0
        assert pr["y"].max() < len(jobs[j]["events"]) and pr["x"].max() <= len(jobs[j]["ref"])


def test_baseline_config_2_full_size(oracle):
    n_reads = 10_000
    pm = sa.Model.load(cases.MODEL_CPG)
    p = sa.default_params()
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    jobs = cases.synthetic_jobs(cases.MODEL_CPG, n_reads, N_EVENTS, cpg_ambiguous=True)
    b = sa.Batch(pm, p, jobs, ambig=amb_p)
    b.run()
    st = b.stats()
    assert st.n_ring_regions == st.n_regions == n_reads and st.n_strip_regions == 0
    _check_probabilities(b, jobs, p.threshold, 37, 0.5)
    for j in range(0, n_reads, 501):
        pr = b.pairs(j)
        assert pr["path"].max() >= 1            # cells with several paths do return pairs of their later paths
        # per event and PATH-summed cell the posteriors stay probabilities
        assert np.bincount(pr["y"], weights=pr["prob_e7"] / 1e7).max() <= 1.0 + 1e-3
    first, per_read = _digest(b, n_reads)
    b.run()
    assert _digest(b, n_reads)[0] == first
    b.close()
    pick = [5, 4242, 9999]
    small = sa.Batch(pm, p, [jobs[j] for j in pick], ambig=amb_p)
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == per_read[j]
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_CPG)
    om = oracle.Model(alpha, k, t10, tab)
    op = cases.oracle_params(oracle, p)
    for q, j in enumerate(pick[:2]):
        exp = cases.oracle_pairs(oracle, om, jobs[j], op, ambig=amb_o)
        worst, n_only = cases.compare_pairs(small.pairs(q), exp, 100, p.threshold)
        assert worst <= 100 and n_only <= 5 and cases.same_order(small.pairs(q), exp)
    small.close()


@pytest.mark.parametrize("threshold", [0.1, 0.01])
def test_baseline_config_3_full_size(oracle, threshold):
    n_reads = 5_000
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    p = sa.default_params(threshold=threshold)
    jobs = cases.hdp_jobs(n_reads, N_EVENTS, table5=pm.table5())
    b = sa.Batch(pm, p, jobs)
    b.run()
    st = b.stats()
    assert st.n_fast_regions == st.n_regions == n_reads
    # the bundled .nhdp's densities are all but identical (sd 15.5 pA for every process): few cells reach 0.1
    _check_probabilities(b, jobs, threshold, 23, 0.01 if threshold >= 0.1 else 5.0)
    first, per_read = _digest(b, n_reads)
    b.run()
    assert _digest(b, n_reads)[0] == first
    b.close()
    # a second batch of the same reads: its page-locked result block is sized from the first one's pairs per event, and at threshold
    # 0.01 (7 GB of pairs) it sweeps in four forward passes so that its copies start early -- the same bytes either way
    b2 = sa.Batch(pm, p, jobs)
    b2.run()
    assert b2.stats().n_chunks == (4 if threshold < 0.05 else 1)
    assert _digest(b2, n_reads)[0] == first
    b2.close()
    if threshold < 0.05:   # ... and as 8-byte records
        b8 = sa.Batch(pm, p, jobs, flags=sa.FLAG_PAIRS8)
        b8.run()
        view, first8 = b8.results_view()
        full = sa.Batch(pm, p, jobs[:40])
        full.run()
        for j in range(40):
            a, c = full.pairs(j), b8.pairs8(j)
            assert np.array_equal(a["x"], c["x"]) and np.array_equal(a["y"], c["y"]) and np.array_equal(a["prob_e7"], c["prob_e7"])
        assert view.shape == (int(first8[-1]), 1) and int(first8[-1]) > 4e8
        b8.close(); full.close()
    pick = [0, 2500, 4999]
    small = sa.Batch(pm, p, [jobs[j] for j in pick])
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == per_read[j]
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_R73)
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(cases.NHDP)
    om.set_to_hdp_expected_values()
    op = cases.oracle_params(oracle, p)
    for q, j in enumerate(pick[:2]):
        exp = cases.oracle_pairs(oracle, om, jobs[j], op)
        worst, n_only = cases.compare_pairs(small.pairs(q), exp, 100, threshold)
        assert worst <= 100 and n_only <= 20
    small.close()


@pytest.mark.parametrize("flavour", ["cpg", "realistic"])
def test_hdp_on_ring_and_strip_kernels_at_bench_size(oracle, flavour):
    """Round 4: HDP emissions on the ring kernels (every CpG cytosine C / E: `bench.py --workload hdp_cpg`) and on the strip kernels
    (anchors of a real guide alignment: `--workload hdp_realistic`), 2000 x 5000-event reads each, both fed by the per-cell-path
    emission plane (k_emit_hdp_ring).  Same size-independent checks as above."""
    n_reads, threshold = 2000, 0.1
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    p = sa.default_params(threshold=threshold)
    jobs = cases.hdp_jobs(n_reads, N_EVENTS, table5=pm.table5())
    amb_p = amb_o = None
    if flavour == "cpg":
        jobs = [dict(j, ref=j["ref"].replace("CG", "XG")) for j in jobs]
        amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    else:
        jobs = cases.thin_anchors_like_a_guide_alignment(jobs)
    b = sa.Batch(pm, p, jobs, ambig=amb_p)
    b.run()
    st = b.stats()
    if flavour == "cpg":
        assert st.n_ring_regions == st.n_regions and st.n_strip_regions == 0
    else:
        assert st.n_strip_regions >= 0.95 * st.n_regions
    _check_probabilities(b, jobs, threshold, 23, 0.01)
    first, per_read = _digest(b, n_reads)
    b.run()
    assert _digest(b, n_reads)[0] == first
    b.close()
    pick = [3, 1999]
    small = sa.Batch(pm, p, [jobs[j] for j in pick], ambig=amb_p)
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == per_read[j]
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_R73)
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(cases.NHDP)
    om.set_to_hdp_expected_values()
    op = cases.oracle_params(oracle, p)
    exp = cases.oracle_pairs(oracle, om, jobs[pick[0]], op, ambig=amb_o)
    worst, n_only = cases.compare_pairs(small.pairs(0), exp, 100, threshold)
    assert worst <= 100 and n_only <= 20
    small.close()
