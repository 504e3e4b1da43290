"""The planner on the device (signalalign_amd/csrc/sa_dplan.inc) against the host planner (sa_plan.c, itself checked
against the CPU restatement's band / split / schedule in tests/test_host_plan.py): every array the kernels read -- regions,
band rows, packed band words, path offsets, k-mer ids, events, traceback segments, checkpoints -- and every total must be
identical byte for byte, so that which planner ran is invisible downstream.  Then the aligned pairs of device-planned
batches against the CPU restatement."""
import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth

import sa_cases as cases

pytestmark = pytest.mark.gpu


def _mixed_jobs():
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 5, 1500, 3)                       # dense anchors
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 2, 5000, 40)                     # several traceback segments
    jobs += cases.realistic_anchor_jobs(cases.MODEL_6MER, 4, 2500, 9)               # sparse anchors: ring kernels
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 3, 3000, 700, thin_anchors=0.35)
    base = synth.make_read(900, 400, alpha, k, tab)
    jobs.append(dict(base, ax=np.zeros(0, dtype=np.int64), ay=np.zeros(0, dtype=np.int64)))       # no anchors at all
    jobs.append(dict(ref=base["ref"][:k + 2], events=base["events"][:4], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0))
    jobs.append(dict(ref=base["ref"][:60], events=base["events"][:0], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0))   # no events
    jobs.append(dict(ref=base["ref"][:k - 1], events=base["events"][:9], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0))  # no k-mers
    jobs.append(synth.make_read(77, 12000, alpha, k, tab))                                                         # 12 segments
    # getAlignedPairsUsingAnchors' ragged-end booleans (sa_job_t.ends): three reads in four with an end that is not ragged
    return [dict(j, ragged=(i & 1, (i >> 1) & 1)) for i, j in enumerate(jobs)]


def test_device_plan_equals_host_plan(monkeypatch):
    pm = sa.Model.load(cases.MODEL_6MER)
    jobs = _mixed_jobs()
    for params in (sa.default_params(), sa.default_params(expansion=20, trace_back=40), sa.default_params(threshold=0.5, expansion=4)):
        assert sa.dplan_compare(pm, params, jobs) == 0
    for env in ({"SA_RING_WIDE": "0"}, {"SA_RING": "0"}, {"SA_F_BUDGET_CELLPATHS": "300000"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        assert sa.dplan_compare(pm, sa.default_params(), jobs) == 0, env
        for k_ in env:
            monkeypatch.delenv(k_)
    pm5 = sa.Model.load(cases.MODEL_5MER)
    assert sa.dplan_compare(pm5, sa.default_params(), cases.synthetic_jobs(cases.MODEL_5MER, 6, 800, 1)) == 0
    hd = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    assert sa.dplan_compare(hd, sa.default_params(), cases.synthetic_jobs(cases.MODEL_R73, 3, 900, 5)) == 0
    # the expectation pass (sa_expect_batch's internal flag): every diagonal keeps all three forward states, no ring kernels
    FLAG_EXPECT_INTERNAL = 0x10000
    assert sa.dplan_compare(pm, sa.default_params(), jobs, flags=FLAG_EXPECT_INTERNAL) == 0
    assert sa.dplan_compare(hd, sa.default_params(threshold=0.1), cases.synthetic_jobs(cases.MODEL_R73, 3, 900, 5),
                            flags=FLAG_EXPECT_INTERNAL) == 0


def test_device_plan_equals_host_plan_with_ambiguous_positions(monkeypatch):
    """Several paths per cell: path offsets, k-mer ids in hdCell_construct2's order, band offsets in cell-paths and the
    per-path neighbour records (compared for every ring-kernel region) -- CpG cytosines C/E, the default table's two- and
    three-letter codes on the R7.3 ACEGOT model, mixed with canonical reads in one batch."""
    p = sa.default_params()
    pmc = sa.Model.load(cases.MODEL_CPG)
    amb = sa.default_ambig({"X": "CE"})
    cpg = cases.synthetic_jobs(cases.MODEL_CPG, 6, 1400, 20, cpg_ambiguous=True) + cases.synthetic_jobs(cases.MODEL_CPG, 2, 900, 3)
    sparse = cases.realistic_anchor_jobs(cases.MODEL_CPG, 2, 1200, 77)
    cpg += [dict(j, ref=j["ref"].replace("CG", "XG")) for j in sparse]
    assert sa.dplan_compare(pmc, p, cpg, ambig=amb) == 0
    monkeypatch.setenv("SA_F_BUDGET_CELLPATHS", "400000")
    assert sa.dplan_compare(pmc, p, cpg, ambig=amb) == 0
    monkeypatch.delenv("SA_F_BUDGET_CELLPATHS")
    pm7 = sa.Model.load(cases.MODEL_R73)
    jobs = []
    for j, job in enumerate(cases.synthetic_jobs(cases.MODEL_R73, 3, 600, 50)):
        ref = list(job["ref"])
        for i in range(7 + j, len(ref) - 6, 23):
            ref[i] = "L" if (i // 23) % 2 == 0 else "P"
        for i in (200, 201, 202):
            ref[i] = "L"
        jobs.append(dict(job, ref="".join(ref)))
    assert sa.dplan_compare(pm7, p, jobs) == 0
    # round 4: the same reads under the HDP model (ring kernels reading the emission plane), and HDP reads with sparse anchors
    hd = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    hd.set_to_hdp_expected_values()
    assert sa.dplan_compare(hd, p, jobs) == 0
    thin = cases.realistic_anchor_jobs(cases.MODEL_R73, 3, 1500, 7)
    assert sa.dplan_compare(hd, p, thin + [dict(thin[0], ref=thin[0]["ref"].replace("CG", "LG"))]) == 0


def test_batches_the_device_planner_leaves_to_the_host(oracle):
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    NOT_TAKEN = 1 << 30
    dense = cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 5)
    assert sa.dplan_compare(pm, p, dense, flags=sa.FLAG_EXACT) == NOT_TAKEN
    assert sa.dplan_compare(pm, sa.default_params(threshold=0.0), dense) == NOT_TAKEN
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    big = synth.make_read(7, 14000, alpha, k, tab)
    hole = (big["ax"] > 1500) & (big["ax"] < 6500)
    big["ax"], big["ay"] = big["ax"][~hole], big["ay"][~hole]
    assert sa.dplan_compare(pm, p, dense + [big]) == NOT_TAKEN                                   # a gap that splits the matrix
    amb = dict(dense[0], ref=dense[0]["ref"][:100] + "R" + dense[0]["ref"][101:])
    assert sa.dplan_compare(pm, p, dense + [amb]) == 0                   # an ambiguity letter (R -> A/G): taken since round 2
    assert sa.dplan_compare(pm, p, dense + [amb], ambig=sa.default_ambig({"R": "AA"})) == NOT_TAKEN   # repeated options
    assert sa.dplan_compare(pm, p, dense + [amb], ambig=sa.default_ambig({"R": "AN"})) == NOT_TAKEN   # option outside the alphabet
    many = dict(dense[0], ref=dense[0]["ref"][:100] + "XXXXXX" + dense[0]["ref"][106:])         # 4^6 paths in one window
    assert sa.dplan_compare(pm, p, dense + [many]) == NOT_TAKEN
    bad = dict(dense[0], ref=dense[0]["ref"][:100] + "N" + dense[0]["ref"][101:])
    assert sa.dplan_compare(pm, p, dense + [bad]) == NOT_TAKEN                                   # (the host planner names the error)
    empty = dict(ref="", events=np.zeros(0), ax=[], ay=[])
    assert sa.dplan_compare(pm, p, dense + [empty]) == NOT_TAKEN
    # ... and such batches still align (host planner), the others through the device planner: same pairs either way
    b = sa.Batch(pm, p, dense + [big])
    b.run()
    op = cases.oracle_params(oracle, p)
    om = oracle.Model(alpha, k, t10, tab)
    exp = cases.oracle_pairs(oracle, om, big, op)
    cases.compare_pairs(b.pairs(3), exp, 100, p.threshold)
    b.close()


def test_device_planned_batches_against_the_oracle(oracle, monkeypatch):
    pm = sa.Model.load(cases.MODEL_6MER)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    p = sa.default_params()
    op = cases.oracle_params(oracle, p)
    jobs = _mixed_jobs()
    b = sa.Batch(pm, p, jobs)
    b.run()
    got = [b.pairs(j) for j in range(len(jobs))]
    b.close()
    for j, job in enumerate(jobs):
        exp = cases.oracle_pairs(oracle, om, job, op)
        cases.compare_pairs(got[j], exp, 100, p.threshold)
        assert cases.same_order(got[j], exp)
    monkeypatch.setenv("SA_DEVICE_PLAN", "0")          # the same batch through the host planner: identical bytes
    b = sa.Batch(pm, p, jobs)
    b.run()
    for j in range(len(jobs)):
        assert np.array_equal(b.pairs(j), got[j]), j
    b.close()
    monkeypatch.delenv("SA_DEVICE_PLAN")
    # ambiguous positions (configs[2] shape): device-planned, against the oracle, and identical to the host-planned run
    pmc = sa.Model.load(cases.MODEL_CPG)
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_CPG)
    omc = oracle.Model(alpha, k, t10, tab)
    amb_p, amb_o = sa.default_ambig({"X": "CE"}), oracle.ambig_map({"X": "CE"})
    cpg = cases.synthetic_jobs(cases.MODEL_CPG, 4, 1100, 20, cpg_ambiguous=True)
    assert sa.dplan_compare(pmc, p, cpg, ambig=amb_p) == 0
    b = sa.Batch(pmc, p, cpg, ambig=amb_p)
    b.run()
    got = [b.pairs(j) for j in range(len(cpg))]
    assert b.stats().n_ring_regions == len(cpg)
    b.close()
    for j, job in enumerate(cpg):
        exp = cases.oracle_pairs(oracle, omc, job, op, ambig=amb_o)
        cases.compare_pairs(got[j], exp, 100, p.threshold)
        assert cases.same_order(got[j], exp)
    monkeypatch.setenv("SA_DEVICE_PLAN", "0")
    b = sa.Batch(pmc, p, cpg, ambig=amb_p)
    b.run()
    for j in range(len(cpg)):
        assert np.array_equal(b.pairs(j), got[j]), j
    b.close()
