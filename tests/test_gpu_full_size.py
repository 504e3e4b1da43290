"""BASELINE configs[1] at its full size (2000 synthetic 5000-event R9.4 reads, band 50, threshold 0.01) -- too large for
the CPU restatement to walk in a test, so what is checked is what the domain guarantees at any size:

  * posteriors are probabilities: every pair within [threshold, 1], and for a fixed event the match posteriors over all
    reference positions sum to 1 at most, up to the reference's own normalisation slack (the event is emitted exactly
    once; the rest of the mass is the insert state);
  * the TSV order (ascending x + y) and coordinates inside the read's matrix;
  * idempotence: a second run of the same batch returns the same bytes (a checksum of per-read checksums);
  * batch-size independence: reads drawn from the 2000 and run alone in a small batch give identical pairs, and for two
    of them the CPU restatement agrees within the 1e-5 bar;
  * the chained maximum-expected-accuracy paths are monotone, use one pair per event, and consist of the read's pairs.
"""
import zlib

import numpy as np
import pytest

import signalalign_amd as sa

import sa_cases as cases

pytestmark = pytest.mark.gpu
N_READS, N_EVENTS = 2000, 5000


def _digest(b, n):
    per_read = []
    for j in range(n):
        per_read.append(zlib.crc32(b.pairs(j).tobytes()))
    return zlib.crc32(np.asarray(per_read, dtype=np.uint32).tobytes()), per_read


def test_baseline_config_1_full_size(oracle):
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, N_READS, N_EVENTS)
    b = sa.Batch(pm, p, jobs)
    b.run()
    total = 0
    for j in range(0, N_READS, 7):                      # every 7th read in full detail
        pr = b.pairs(j)
        total += len(pr)
        assert len(pr) > 0.5 * len(jobs[j]["events"])
        assert pr["prob_e7"].min() >= int(p.threshold * 1e7) and pr["prob_e7"].max() <= 10_000_000
        assert np.all(np.diff(pr["x"] + pr["y"]) >= 0)
        assert pr["x"].min() >= 0 and pr["y"].min() >= 0
        assert pr["y"].max() < len(jobs[j]["events"]) and pr["x"].max() <= len(jobs[j]["ref"])
        per_event = np.bincount(pr["y"], weights=pr["prob_e7"] / 1e7)
        # not exactly 1: the reference divides by a total probability it refreshes every 10th diagonal only, and inside
        # a band the diagonals' totals differ by what leaks out of the band (measured: up to 7e-5 above 1)
        assert per_event.max() <= 1.0 + 1e-3
    assert total > 0
    first, per_read = _digest(b, N_READS)
    mea = b.mea()
    b.run()                                             # idempotence
    again, _ = _digest(b, N_READS)
    assert again == first

    rng = np.random.default_rng(0)
    pick = sorted(rng.choice(N_READS, 6, replace=False).tolist())
    small = sa.Batch(pm, p, [jobs[j] for j in pick])
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == per_read[j]       # same bytes alone as among 2000
    small.close()
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    op = cases.oracle_params(oracle, p)
    for j in pick[:2]:
        exp = cases.oracle_pairs(oracle, om, jobs[j], op)
        worst, n_only = cases.compare_pairs(b.pairs(j), exp, 100, p.threshold)   # rows near the threshold may differ
        assert worst <= 100 and n_only <= 5

    for j in range(0, N_READS, 11):
        path, best, st = mea[j]
        pr = b.pairs(j)
        assert st == 0 and len(path) > 0.5 * len(jobs[j]["events"])
        assert np.all(np.diff(path[:, 0]) >= 0) and np.all(np.diff(path[:, 1]) > 0)
        have = set(zip(pr["x"].tolist(), pr["y"].tolist()))
        assert all((int(x), int(y)) in have for x, y in path[:: 17])
        assert best <= len(path) + 1e-9
    b.close()


def test_realistic_anchor_density_full_size(oracle, monkeypatch):
    """The same reads with the anchors a real guide alignment leaves (bands of 100-300 cells) at bench size, 5000 events a
    read: every region runs on the strip kernels.  Checked: posteriors are probabilities in TSV order; a second run and a
    second batch give the same bytes (seam arrays, side buffers and LDS maxima leave nothing behind); reads run alone in a
    small batch give the same bytes; the ring kernels (SA_STRIP=0) give the same bytes for every read -- they fold a cell's
    terms in the same order from LDS rows instead of registers and seams --; two reads agree with the CPU restatement."""
    n_reads = 600
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.realistic_anchor_jobs(cases.MODEL_6MER, n_reads, N_EVENTS)
    b = sa.Batch(pm, p, jobs)
    b.run()
    st = b.stats()
    assert st.n_strip_regions == st.n_regions == n_reads
    for j in range(0, n_reads, 5):
        pr = b.pairs(j)
        assert len(pr) > 0.5 * len(jobs[j]["events"])
        assert pr["prob_e7"].min() >= int(p.threshold * 1e7) and pr["prob_e7"].max() <= 10_000_000
        assert np.all(np.diff(pr["x"] + pr["y"]) >= 0)
        assert pr["y"].max() < len(jobs[j]["events"]) and pr["x"].max() <= len(jobs[j]["ref"])
        assert np.bincount(pr["y"], weights=pr["prob_e7"] / 1e7).max() <= 1.0 + 1e-3
    first, per_read = _digest(b, n_reads)
    b.run()
    assert _digest(b, n_reads)[0] == first
    b.close()
    b2 = sa.Batch(pm, p, jobs)
    b2.run()
    assert _digest(b2, n_reads)[0] == first
    b2.close()
    pick = [3, 77, 311, 598]
    small = sa.Batch(pm, p, [jobs[j] for j in pick])
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == per_read[j]
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    op = cases.oracle_params(oracle, p)
    for q, j in enumerate(pick[:2]):
        exp = cases.oracle_pairs(oracle, om, jobs[j], op)
        worst, n_only = cases.compare_pairs(small.pairs(q), exp, 100, p.threshold)
        assert worst <= 10 and n_only <= 5 and cases.same_order(small.pairs(q), exp)
    small.close()
    monkeypatch.setenv("SA_STRIP", "0")
    ring = sa.Batch(pm, p, jobs)
    ring.run()
    assert ring.stats().n_strip_regions == 0
    assert _digest(ring, n_reads)[1] == per_read
    ring.close()
