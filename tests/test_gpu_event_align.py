"""Event <-> k-mer pre-alignment (adaptive_banded_simple_event_align, impl/eventAligner.c:899-1235): the HIP path
against the CPU restatement.  PARITY UNPINNED against the reference itself -- its tests of this function read fast5
files (tests/eventAlignerTests.c:223-320, :404-430) -- so what is checked here is: bit-identical pair lists and status
words between the two restatements, and properties any correct alignment has (monotone, spanning, every event between
the first and last aligned event used once, agreement with the generator's ground truth).
"""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def _jobs(oracle, model_path, n, n_events, first=0, noise=1.0):
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    om = oracle.Model(alpha, k, t10, tab)
    pm = sa.Model.load(model_path)
    jobs, truth = [], []
    for i in range(n):
        r = synth.make_read(first + i, n_events, alpha, k, tab)
        ev = np.ascontiguousarray(np.asarray(r["events4"])[:, 0])
        if noise != 1.0:
            rng = np.random.default_rng(i)
            ev = ev + rng.normal(0, noise, len(ev))
        seq = r["ref"]
        sh, sc = sa.scalings_mom(pm, seq, ev)
        osh, osc = oracle.scalings_mom(om, ev, oracle.kmer_ids_of(om, seq))
        assert sh == osh and sc == osc
        jobs.append(dict(sequence=seq, event_mean=ev, scale=sc, shift=sh, var=1.0))
        truth.append(r)
    return pm, om, jobs, truth


@pytest.mark.parametrize("model_path,n_events", [(cases.MODEL_6MER, 1500), (cases.MODEL_5MER, 600), (cases.MODEL_6MER, 40)])
def test_matches_the_cpu_restatement_bit_for_bit(oracle, model_path, n_events):
    pm, om, jobs, truth = _jobs(oracle, model_path, 6, n_events, first=77)
    got = sa.event_align_batch(pm, jobs)
    for j, job in enumerate(jobs):
        om.set_read_params(job["scale"], job["shift"], 1.0)
        ek, ee, est = oracle.event_align(om, job["event_mean"], oracle.kmer_ids_of(om, job["sequence"]))
        gk, ge, gst = got[j]
        assert gst == est, j
        assert np.array_equal(gk, ek) and np.array_equal(ge, ee), j


def test_alignment_properties_and_ground_truth(oracle):
    pm, om, jobs, truth = _jobs(oracle, cases.MODEL_6MER, 4, 3000, first=500)
    got = sa.event_align_batch(pm, jobs)
    for (gk, ge, st), job, r in zip(got, jobs, truth):
        assert st == 0 and len(gk) > 0
        n_kmers = len(job["sequence"]) - 5
        assert gk[0] == 0 and gk[-1] == n_kmers - 1                       # spanned
        dk, de = np.diff(gk), np.diff(ge)
        assert ((dk == 0) | (dk == 1)).all() and ((de == 0) | (de == 1)).all() and ((dk + de) >= 1).all()
        assert len(set(zip(gk.tolist(), ge.tolist()))) == len(gk)
        # the generator knows which k-mer produced every event: the Viterbi path finds it within one position almost always
        emap = np.asarray(r["event_map"])                                 # base -> first event
        owner = np.searchsorted(emap[:n_kmers], ge, side="right") - 1
        assert (np.abs(owner - gk) <= 1).mean() > 0.97


def test_rejected_alignment_reports_why(oracle):
    # events that have nothing to do with the sequence: the average emission check fires and the list comes back empty
    pm, om, jobs, truth = _jobs(oracle, cases.MODEL_6MER, 2, 800, first=900)
    rng = np.random.default_rng(5)
    jobs[0]["event_mean"] = rng.uniform(40, 140, len(jobs[0]["event_mean"]))
    got = sa.event_align_batch(pm, jobs)
    om.set_read_params(jobs[0]["scale"], jobs[0]["shift"], 1.0)
    ek, ee, est = oracle.event_align(om, jobs[0]["event_mean"], oracle.kmer_ids_of(om, jobs[0]["sequence"]))
    assert got[0][2] == est and est != 0 and len(got[0][0]) == 0 and len(ek) == 0
    assert got[1][2] == 0 and len(got[1][0]) > 0


def test_rna_kmer_list(oracle):
    # build_kmer_list(..., rna=true): U -> T, every k-mer reversed (impl/eventAligner.c:772-780)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_5MER)
    om = oracle.Model(alpha, k, t10, tab)
    pm = sa.Model.load(cases.MODEL_5MER)
    r = synth.make_read(4242, 700, alpha, k, tab)
    seq = r["ref"].replace("T", "U")
    ids = oracle.kmer_ids_of(om, seq, rna=True)
    assert not np.array_equal(ids, oracle.kmer_ids_of(om, r["ref"]))
    # events drawn for the reversed k-mers, so that the alignment is accepted
    rng = np.random.default_rng(3)
    counts = rng.integers(1, 3, size=len(ids))
    owner = np.repeat(np.arange(len(ids)), counts)
    ev = rng.normal(tab[5 * ids[owner]], tab[5 * ids[owner] + 1])
    sh, sc = sa.scalings_mom(pm, seq, ev, flags=sa.FLAG_RNA)
    assert (sh, sc) == oracle.scalings_mom(om, ev, ids)
    got = sa.event_align_batch(pm, [dict(sequence=seq, event_mean=ev, scale=sc, shift=sh, var=1.0)], flags=sa.FLAG_RNA)[0]
    om.set_read_params(sc, sh, 1.0)
    ek, ee, est = oracle.event_align(om, ev, ids)
    assert got[2] == est == 0 and np.array_equal(got[0], ek) and np.array_equal(got[1], ee)


def test_non_unit_variance_and_a_long_read(oracle):
    """var != 1 takes the kernel instance that keeps the division by var (the reference always passes 1, but the
    arithmetic is part of the emission); a 12000-event read crosses many refills of the event and k-mer stream buffers
    and many 64-band traceback blocks."""
    pm, om, jobs, truth = _jobs(oracle, cases.MODEL_6MER, 3, 2500, first=31)
    for j, v in zip(jobs, (1.3, 0.8, 1.0)):
        j["var"] = v
    long_pm, long_om, long_jobs, _ = _jobs(oracle, cases.MODEL_6MER, 1, 12000, first=5)
    jobs = jobs + long_jobs
    got = sa.event_align_batch(pm, jobs)
    for j, job in enumerate(jobs):
        om.set_read_params(job["scale"], job["shift"], job["var"])
        ek, ee, est = oracle.event_align(om, job["event_mean"], oracle.kmer_ids_of(om, job["sequence"]))
        gk, ge, gst = got[j]
        assert gst == est, j
        assert np.array_equal(gk, ek) and np.array_equal(ge, ee), j
    assert got[3][2] == 0 and len(got[3][0]) > 12000
    # all-unit batch again right after: the other kernel instance, same workspace
    again = sa.event_align_batch(pm, long_jobs)
    assert np.array_equal(again[0][0], got[3][0]) and np.array_equal(again[0][1], got[3][1])
