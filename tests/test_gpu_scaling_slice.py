"""BASELINE configs[4] (100k synthetic 10k-event reads sharded over 8 GPUs) -- the single-GPU slice at real size.

One GPU's share of that job does not fit the forward storage of one pass (24 B per band cell, 60 % of free HBM), so the
planner cuts it into several passes; nothing here uses the SA_F_BUDGET_CELLPATHS test hook.  The run is too large for the
CPU restatement to walk, so it is checked through what holds at any size: idempotence (a checksum of per-read
checksums over two runs), batch-size independence (reads re-run alone give the same bytes -- including reads on both
sides of a pass boundary), probabilities within [threshold, 1], TSV order -- and two reads against the CPU restatement
within the 1e-5 bar.  The read-to-rank partition of the full job is src/signalalign/signalAlignment.py:694-737's
one-process-per-read made explicit (signalalign_amd/shard.py, tests/test_multi_rank.py).
"""
import zlib

import numpy as np
import pytest

import signalalign_amd as sa

import sa_cases as cases

pytestmark = pytest.mark.gpu
N_READS, N_EVENTS = 11000, 10000    # 11000 x 7.9e5 band cells x 24 B = 208 GB of forward storage: more than one pass
                                    # (9000 reads still fit one: measured on the 288 GB card)


def test_ten_k_event_reads_need_several_forward_storage_passes(oracle):
    pm = sa.Model.load(cases.MODEL_6MER)
    p = sa.default_params()
    jobs = cases.synthetic_jobs(cases.MODEL_6MER, N_READS, N_EVENTS, first_index=500000)
    b = sa.Batch(pm, p, jobs)
    st = b.stats()
    assert st.n_chunks >= 2, "expected more than one forward-storage pass, got %d" % st.n_chunks
    assert st.n_fast_regions == st.n_regions == N_READS
    b.run()
    sample = list(range(0, N_READS, 9))
    crc = {j: zlib.crc32(b.pairs(j).tobytes()) for j in sample}
    for j in sample[::10]:
        pr = b.pairs(j)
        assert len(pr) > 0.5 * len(jobs[j]["events"])
        assert pr["prob_e7"].min() >= int(p.threshold * 1e7) and pr["prob_e7"].max() <= 10_000_000
        assert np.all(np.diff(pr["x"] + pr["y"]) >= 0)
        assert pr["y"].max() < len(jobs[j]["events"]) and pr["x"].max() <= len(jobs[j]["ref"])
    first = zlib.crc32(np.asarray([crc[j] for j in sample], dtype=np.uint32).tobytes())
    b.run()                                                     # idempotence across the passes
    again = zlib.crc32(np.asarray([zlib.crc32(b.pairs(j).tobytes()) for j in sample], dtype=np.uint32).tobytes())
    assert again == first
    # the same reads alone (one pass): identical bytes; the picks straddle the whole list, so every pass is represented
    pick = [sample[0], sample[len(sample) // 3], sample[len(sample) // 2], sample[2 * len(sample) // 3], sample[-1]]
    small = sa.Batch(pm, p, [jobs[j] for j in pick])
    assert small.stats().n_chunks == 1
    small.run()
    for q, j in enumerate(pick):
        assert zlib.crc32(small.pairs(q).tobytes()) == crc[j], j
    small.close()
    alpha, k, t10, tab = sa.synth.parse_model_table(cases.MODEL_6MER)
    om = oracle.Model(alpha, k, t10, tab)
    op = cases.oracle_params(oracle, p)
    for j in (pick[0], pick[-1]):
        exp = cases.oracle_pairs(oracle, om, jobs[j], op)
        worst, n_only = cases.compare_pairs(b.pairs(j), exp, 100, p.threshold)
        assert worst <= 100 and n_only <= 5
        assert cases.same_order(b.pairs(j), exp)
    b.close()
