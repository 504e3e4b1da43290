"""SA_FLAG_INPUTS_IN_HOST_BLOCK (include/signalalign_hip.h): the reads' event records and anchor arrays stay in the caller's
page-locked block, cross PCIe with one DMA and are checked, narrowed and gathered by a kernel (k_dplan_ingest,
signalalign_amd/csrc/sa_dplan.inc) instead of by host cores.  Nothing downstream may tell: the planner's arrays, the aligned
pairs and the expectations are identical bytes with and without the flag, and reads the device checks turn down end where
they end without it (host planner: same pairs or the same named error)."""
import ctypes as C

import numpy as np
import pytest

import signalalign_amd as sa
from signalalign_amd import synth
from signalalign_amd._capi import jobs_bytes_in_block

import sa_cases as cases

pytestmark = pytest.mark.gpu

FLAG = sa.FLAG_INPUTS_IN_HOST_BLOCK


def _records(job, width=4):
    """The reference's event records (mean, sd, noise, duration per event: nanopore.c's NB_EVENT_PARAMS doubles)."""
    ev = np.asarray(job["events"], dtype=np.float64)
    if ev.ndim == 2:
        return job
    rec = np.zeros((len(ev), width))
    rec[:, 0] = ev
    rec[:, 1] = 1.0
    rec[:, 2] = 1.3e-3
    rec[:, 3] = np.arange(len(ev)) * 1e-3
    return dict(job, events=rec)


def _jobs():
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 5, 1500, 3)
    jobs += cases.synthetic_jobs(cases.MODEL_6MER, 2, 5000, 40)
    jobs += cases.realistic_anchor_jobs(cases.MODEL_6MER, 4, 2500, 9)
    base = synth.make_read(900, 400, alpha, k, tab)
    jobs.append(dict(base, ax=np.zeros(0, dtype=np.int64), ay=np.zeros(0, dtype=np.int64)))       # no anchors at all
    jobs.append(dict(ref=base["ref"][:60], events=base["events"][:0], ax=[], ay=[], scale=1.0, shift=0.0, var=1.0))   # no events
    jobs.append(synth.make_read(77, 12000, alpha, k, tab))
    return jobs


@pytest.mark.parametrize("records,interleaved", [(False, False), (True, False), (True, True)])
def test_same_plan_and_same_pairs_with_the_inputs_left_in_the_callers_block(records, interleaved):
    """interleaved: the block holds read after read (one DMA ahead of the planning kernels); otherwise all event records, then
    all anchors -- the anchors travel first and the event records on the batch's own stream, behind sa_batch_create."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = [_records(j) for j in _jobs()] if records else _jobs()
    ja = sa.JobArray(jobs, host_block=True, interleaved=interleaved)
    assert sa.dplan_compare(pm, p, ja, flags=FLAG) == 0        # events, band rows, segments, ...: the host planner's bytes
    ref = sa.Batch(pm, p, jobs)
    ref.run()
    b = sa.Batch(pm, p, ja, flags=FLAG)
    b.run()
    assert b.stats().n_fast_regions == ref.stats().n_fast_regions and b.stats().n_ring_regions == ref.stats().n_ring_regions
    for j in range(len(jobs)):
        assert np.array_equal(b.pairs(j), ref.pairs(j)), j
    b.close()
    # the streaming form: created in two halves, run on the library's thread
    b = sa.Batch(pm, p, ja, flags=FLAG, deferred=True)
    b.start()
    b.wait()
    for j in range(len(jobs)):
        assert np.array_equal(b.pairs(j), ref.pairs(j)), j
    b.close()
    ref.close()


def test_a_batch_that_uses_a_small_part_of_a_shared_arena(monkeypatch):
    """One page-locked arena holds the arrays of many reads and a batch aligns every fourth of them: the covering range of its
    arrays is four times what it names.  The library then packs the batch with host threads (as without the flag) instead of
    sending -- and holding in HBM -- the other reads' bytes: same pairs, and the batch's device_bytes do not grow by the arena."""
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = [_records(j) for j in cases.synthetic_jobs(cases.MODEL_6MER, 16, 9000, 500)]   # (2.2 MB named of 8.6 MB covered)
    ja = sa.JobArray(jobs, host_block=True)
    some = [ja.jobs[i] for i in range(0, len(jobs), 4)]
    ref = sa.Batch(pm, p, [jobs[i] for i in range(0, len(jobs), 4)])
    ref.run()
    b = sa.Batch(pm, p, some, flags=FLAG)
    b.run()
    for j in range(len(some)):
        assert np.array_equal(b.pairs(j), ref.pairs(j)), j
    assert b.stats().device_bytes <= ref.stats().device_bytes * 1.02 + 4096
    b.close()
    # all of them: the block travels as it is, and its image in HBM is counted
    full = sa.Batch(pm, p, ja, flags=FLAG)
    full_ref = sa.Batch(pm, p, jobs)
    assert full.stats().device_bytes >= full_ref.stats().device_bytes + 0.9 * jobs_bytes_in_block(jobs) - 4096
    full.close()
    full_ref.close()
    ref.close()


def test_reads_the_device_checks_turn_down(oracle):
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    dense = cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 5)
    NOT_TAKEN = 1 << 30
    # a gap between anchors that splits the matrix: host planner, same pairs as without the flag
    big = synth.make_read(7, 14000, alpha, k, tab)
    hole = (big["ax"] > 1500) & (big["ax"] < 6500)
    big["ax"], big["ay"] = big["ax"][~hole], big["ay"][~hole]
    ja = sa.JobArray(dense + [big], host_block=True)
    assert sa.dplan_compare(pm, p, ja, flags=FLAG) == NOT_TAKEN
    b, ref = sa.Batch(pm, p, ja, flags=FLAG), sa.Batch(pm, p, dense + [big])
    b.run()
    ref.run()
    for j in range(4):
        assert np.array_equal(b.pairs(j), ref.pairs(j)), j
    b.close()
    ref.close()
    # anchors that run backwards, leave the matrix or repeat: what sa_batch_create answers without the flag
    for spoil in ("backwards", "outside", "repeat", "negative"):
        bad = dict(dense[1], ax=np.array(dense[1]["ax"]), ay=np.array(dense[1]["ay"]))
        if spoil == "backwards":
            bad["ax"][40], bad["ax"][41] = bad["ax"][41], bad["ax"][40]
        elif spoil == "outside":
            bad["ay"][-1] = len(bad["events"]) + 5
        elif spoil == "repeat":
            bad["ay"][100] = bad["ay"][99]
        else:
            bad["ax"][0] = -1
        want = None
        try:
            sa.Batch(pm, p, [dense[0], bad]).close()
        except sa.SaError as e:
            want = e.code
        got = None
        try:
            sa.Batch(pm, p, sa.JobArray([dense[0], bad], host_block=True), flags=FLAG).close()
        except sa.SaError as e:
            got = e.code
        assert got == want and want is not None, (spoil, got, want)


def test_pointers_outside_a_block_are_refused():
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, 3, 900, 5)
    with pytest.raises(sa.SaError) as e:
        sa.Batch(pm, p, jobs, flags=FLAG)                      # numpy's own memory
    assert e.value.code == -1                                  # SA_EINVAL
    # a read whose events run past the end of the block
    ja = sa.JobArray(jobs, host_block=True)
    ja.arr[2].n_events = ja.block.nbytes // 8 + 1
    with pytest.raises(sa.SaError):
        sa.Batch(pm, p, ja, flags=FLAG)
    # batches the host plans read the same memory: the flag changes nothing
    ja = sa.JobArray(jobs, host_block=True)
    b, ref = sa.Batch(pm, p, ja, flags=FLAG | sa.FLAG_EXACT), sa.Batch(pm, p, jobs, flags=sa.FLAG_EXACT)
    b.run()
    ref.run()
    for j in range(3):
        assert np.array_equal(b.pairs(j), ref.pairs(j))
    b.close()
    ref.close()
    # blocks come and go
    L = sa.lib()
    blk = [L.sa_host_alloc(1 << 20) for _ in range(4)]
    assert all(blk) and len(set(blk)) == 4
    for q in blk:
        L.sa_host_free(q)
    L.sa_host_free(None)
    L.sa_host_free(C.c_void_p(12345))                          # not a block: ignored


def test_hdp_and_expectations_with_the_flag():
    pm = sa.Model.load(cases.MODEL_R73, cases.NHDP)
    pm.set_to_hdp_expected_values()
    p = sa.default_params(threshold=0.1)
    jobs = [_records(j) for j in cases.hdp_jobs(6, 1200, table5=pm.table5())]
    ja = sa.JobArray(jobs, host_block=True)
    b, ref = sa.Batch(pm, p, ja, flags=FLAG), sa.Batch(pm, p, jobs)
    b.run()
    ref.run()
    for j in range(len(jobs)):
        assert np.array_equal(b.pairs(j), ref.pairs(j)), j
    b.close()
    ref.close()
    g = sa.Model.load(cases.MODEL_6MER)
    pg = sa.default_params()
    jobs = [_records(j) for j in cases.synthetic_jobs(cases.MODEL_6MER, 5, 1500, 3)]
    a = sa.expect_batch(g, pg, jobs)
    c = sa.expect_batch(g, pg, sa.JobArray(jobs, host_block=True), flags=FLAG)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
