"""Randomised parity: many small jobs of random shape against the CPU oracle.

Shapes drawn per job (fixed seed): model (5-mer / 6-mer ACGT / CpG ACEGT with ambiguous positions), read length from a
handful of events to a few thousand (so that single-traceback, multi-traceback and split matrices all occur), anchors
dropped at random (from dense to none at all), several parameter sets (band expansion, traceback overlap, threshold).
SA_FLAG_EXACT must reproduce the oracle bit for bit; the default kernels must stay within the 1e-5 bar and keep the
row order.  The reference's own tests cover these regimes one read at a time; this walks the product
of them.
"""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def _jobs_for(model_path, rng, n, ambiguous):
    alpha, k, t10, tab = synth.parse_model_table(model_path)
    jobs = []
    for i in range(n):
        n_events = int(rng.choice([3, 9, 25, 60, 150, 400, 900, 1700, 2600]))
        # (ambiguous reads: every CpG cytosine, or only every 5th / 20th -- sparse variant positions take the ring kernels' other instance)
        every = int(rng.choice([1, 1, 5, 20])) if ambiguous else 1
        job = synth.make_read(int(rng.integers(0, 10 ** 6)), n_events, alpha, k, tab, cpg_ambiguous=ambiguous, cpg_every=every)
        keep = rng.random()
        if keep < 0.15:
            m = np.zeros(len(job["ax"]), dtype=bool)            # no anchors at all: the whole matrix is one band
        elif keep < 0.6:
            m = rng.random(len(job["ax"])) < rng.uniform(0.02, 0.9)  # thinned at random
        else:
            m = np.ones(len(job["ax"]), dtype=bool)
        if n_events > 1200 and rng.random() < 0.5 and len(job["ax"]) > 40:
            lo = int(rng.integers(5, len(job["ax"]) // 2))       # a long anchor-free stretch
            m[lo:lo + int(rng.integers(20, len(job["ax"]) // 3))] = False
        job["ax"], job["ay"] = job["ax"][m], job["ay"][m]
        # getAlignedPairsUsingAnchors' two ragged-end booleans (sa_job_t.ends): every combination, (1, 1) -- signalMachine's -- for
        # one job in four (a function of the job's index: the seeds' reads stay what they were)
        job["ragged"] = (i & 1, (i >> 1) & 1)
        jobs.append(job)
    return alpha, k, t10, tab, jobs


@pytest.mark.parametrize("model_path,ambiguous,seed", [(cases.MODEL_6MER, False, 11), (cases.MODEL_5MER, False, 12),
                                                        (cases.MODEL_CPG, True, 13)])
def test_random_shapes(oracle, model_path, ambiguous, seed, require_strips=True):
    rng = np.random.default_rng(seed)
    alpha, k, t10, tab, jobs = _jobs_for(model_path, rng, 14, ambiguous)
    pm = sa.Model.load(model_path)
    om = oracle.Model(alpha, k, t10, tab)
    amb_p = sa.default_ambig({"X": "CE"}) if ambiguous else None
    amb_o = oracle.ambig_map({"X": "CE"}) if ambiguous else None
    strips_seen = 0
    for expansion, trace_back, threshold, split in ((50, 100, 0.01, 3000 * 3000), (20, 30, 0.2, 3000 * 3000),
                                                    (50, 100, 0.01, 250 * 250)):
        p = sa.default_params(threshold=threshold, expansion=expansion, trace_back=trace_back, split=split)
        op = cases.oracle_params(oracle, p)
        exp = [cases.oracle_pairs(oracle, om, job, op, ambig=amb_o) for job in jobs]
        for flags in (sa.FLAG_EXACT, 0, sa.FLAG_FORCE_GENERIC):
            b = sa.Batch(pm, p, jobs, ambig=amb_p, flags=flags)
            b.run()
            strips_seen += b.stats().n_strip_regions
            for j in range(len(jobs)):
                got = b.pairs(j)
                if flags == sa.FLAG_EXACT:
                    assert len(got) == len(exp[j]), (j, expansion, len(jobs[j]["events"]))
                    for f in ("x", "y", "path", "kmer_id", "prob_e7"):
                        assert np.array_equal(got[f], exp[j][f]), (j, f, expansion)
                else:
                    cases.compare_pairs(got, exp[j], 100, p.threshold)
                    assert cases.same_order(got, exp[j])
            b.close()
        # the same reads left in a page-locked block of the caller (SA_FLAG_INPUTS_IN_HOST_BLOCK): checked and packed on the
        # device, or -- split regions, odd anchors -- handed back to the host planner: the same bytes as the plain batch
        ref_b = sa.Batch(pm, p, jobs, ambig=amb_p)
        ref_b.run()
        ja = sa.JobArray(jobs, host_block=True, interleaved=bool(seed & 1))
        blk = sa.Batch(pm, p, ja, ambig=amb_p, flags=sa.FLAG_INPUTS_IN_HOST_BLOCK)
        blk.run()
        for j in range(len(jobs)):
            assert np.array_equal(blk.pairs(j), ref_b.pairs(j)), (j, expansion, split)
        blk.close()
        # round 5, the two result-side flags under the same shapes: with ambiguity letters only the rows whose reference k-mer holds
        # an X leave the device (SA_FLAG_VC_ROWS), the others are counted and summed; without, the 8-byte records hold the same pairs
        if ambiguous:
            vc = sa.Batch(pm, p, jobs, ambig=amb_p, flags=sa.FLAG_VC_ROWS)
            vc.run()
            for j, job in enumerate(jobs):
                a = ref_b.pairs(j)
                keep = np.array([("X" in job["ref"][x:x + k]) for x in a["x"]], dtype=bool)
                assert np.array_equal(a[keep], vc.pairs(j)), (j, expansion, split)
                assert vc.all_pairs_summary(j) == (len(a), int(a["prob_e7"].sum()))
            vc.close()
        else:
            p8 = sa.Batch(pm, p, jobs, flags=sa.FLAG_PAIRS8)
            p8.run()
            for j in range(len(jobs)):
                a, c = ref_b.pairs(j), p8.pairs8(j)
                assert len(a) == len(c) and np.array_equal(a["x"], c["x"]) and np.array_equal(a["y"], c["y"]) \
                    and np.array_equal(a["prob_e7"], c["prob_e7"]), (j, expansion, split)
            p8.close()
        ref_b.close()
    # thinned and absent anchors leave wide one-path bands: those regions run on the strip kernels (sa_strip.inc), small
    # split rectangles and ragged ends included
    # (with the suite's seeds; probes/fuzz_campaign.py walks other seeds, whose reads need not include such a region)
    assert ambiguous or strips_seen > 0 or not require_strips


def test_random_shapes_hdp_and_expectations(oracle, seed0=31):
    """The round-3 paths under the same random shapes: HDP emissions through the emission plane (k_emit_hdp) and the register
    kernels -- including split matrices (several regions per read, each with its own plane), reads of a few events and
    anchor-free matrices whose wide diagonals take the in-kernel memory-resident stretches -- against the oracle at the HDP bar;
    and the expectation pass on the register kernels against the memory-resident checker and the oracle."""
    rng = np.random.default_rng(seed0)
    alpha, k, t10, tab, jobs = _jobs_for(cases.MODEL_R73, rng, 10, False)
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(cases.NHDP)
    om.set_to_hdp_expected_values()
    for expansion, trace_back, threshold, split in ((50, 100, 0.1, 3000 * 3000), (20, 30, 0.05, 250 * 250)):
        p = sa.default_params(threshold=threshold, expansion=expansion, trace_back=trace_back, split=split)
        op = cases.oracle_params(oracle, p)
        b = sa.Batch(pm, p, jobs)
        b.run()
        st = b.stats()
        # (round 4: HDP regions whose band is mostly wider than a wave take the strip kernels, the rest the register kernels; nothing
        # is left to the memory-resident ones.  The small limit splits / drops matrices)
        assert st.n_fast_regions + st.n_ring_regions == st.n_regions and (split > 10 ** 6 or st.n_regions != len(jobs) or seed0 != 31)
        for j, job in enumerate(jobs):
            cases.compare_pairs(b.pairs(j), cases.oracle_pairs(oracle, om, job, op), 100, p.threshold)
        b.close()
    # expectation pass, Gaussian model, random shapes: register kernels == checker (1e-9), both == oracle
    rng = np.random.default_rng(seed0 + 1)
    alpha, k, t10, tab, jobs = _jobs_for(cases.MODEL_6MER, rng, 10, False)
    pg = sa.Model.load(cases.MODEL_6MER)
    og = oracle.Model(alpha, k, t10, tab)
    for expansion, trace_back, split in ((50, 100, 3000 * 3000), (20, 30, 250 * 250)):
        p = sa.default_params(threshold=0.01, expansion=expansion, trace_back=trace_back, split=split)
        op = cases.oracle_params(oracle, p)
        ft, fl, _ = sa.expect_batch(pg, p, jobs)
        gt, gl, _ = sa.expect_batch(pg, p, jobs, flags=sa.FLAG_FORCE_GENERIC)
        for j, job in enumerate(jobs):
            og.set_read_params(job["scale"], job["shift"], job["var"])
            t, l, _, _, _ = oracle.expectations(og, job["ref"], job["events"], job["ax"], job["ay"], op, ragged=job["ragged"])
            np.testing.assert_allclose(ft[j], t, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(gt[j], t, rtol=1e-9, atol=1e-10)
            assert abs(fl[j] - l) <= 1e-12 * max(abs(l), 1.0) and abs(gl[j] - l) <= 1e-12 * max(abs(l), 1.0)

    # round 4: the same two passes with ambiguity letters (every, every 5th or every 20th CpG cytosine C / E) -- HDP emissions from
    # the per-cell-path plane on the ring kernels, and the ring kernels' expectation variant against checker and oracle
    rng = np.random.default_rng(seed0 + 2)
    alpha, k, t10, tab, jobs = _jobs_for(cases.MODEL_R73, rng, 8, True)
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    om = oracle.Model(alpha, k, t10, tab)
    om.load_hdp(cases.NHDP)
    om.set_to_hdp_expected_values()
    for expansion, trace_back, threshold, split in ((50, 100, 0.1, 3000 * 3000), (20, 30, 0.05, 250 * 250)):
        p = sa.default_params(threshold=threshold, expansion=expansion, trace_back=trace_back, split=split)
        op = cases.oracle_params(oracle, p)
        b = sa.Batch(pm, p, jobs, ambig=amb_p)
        b.run()
        for j, job in enumerate(jobs):
            cases.compare_pairs(b.pairs(j), cases.oracle_pairs(oracle, om, job, op, ambig=amb_o), 100, p.threshold)
        b.close()
    rng = np.random.default_rng(seed0 + 3)
    alpha, k, t10, tab, jobs = _jobs_for(cases.MODEL_CPG, rng, 8, True)
    pc = sa.Model.load(cases.MODEL_CPG)
    oc = oracle.Model(alpha, k, t10, tab)
    for expansion, trace_back, split in ((50, 100, 3000 * 3000), (20, 30, 250 * 250)):
        p = sa.default_params(threshold=0.01, expansion=expansion, trace_back=trace_back, split=split)
        op = cases.oracle_params(oracle, p)
        ft, fl, _ = sa.expect_batch(pc, p, jobs, ambig=amb_p)
        gt, gl, _ = sa.expect_batch(pc, p, jobs, ambig=amb_p, flags=sa.FLAG_FORCE_GENERIC)
        for j, job in enumerate(jobs):
            oc.set_read_params(job["scale"], job["shift"], job["var"])
            t, l, _, _, _ = oracle.expectations(oc, job["ref"], job["events"], job["ax"], job["ay"], op, ambig=amb_o, ragged=job["ragged"])
            np.testing.assert_allclose(ft[j], t, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(gt[j], t, rtol=1e-9, atol=1e-10)
            assert abs(fl[j] - l) <= 1e-12 * max(abs(l), 1.0) and abs(gl[j] - l) <= 1e-12 * max(abs(l), 1.0)
