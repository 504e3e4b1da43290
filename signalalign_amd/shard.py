"""Read sharding across the GPUs of one node.

Reads are independent (the reference runs one process per read, src/signalalign/signalAlignment.py:694-737), so
multi-GPU execution is a partition of the read list with no collective on the data path: each rank aligns its
shard, results are concatenated in read order on the host.  This module only decides who aligns what.
"""
import numpy as np


def shard_indices(costs, rank, world):
    """Deal reads to ranks by descending cost, always to the currently least-loaded rank (longest-processing-time
    first).  Deterministic; every read goes to exactly one rank.  costs: per-read work estimate (e.g. events)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world, dtype=np.float64)
    owner = np.empty(len(costs), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += costs[i]
    return np.nonzero(owner == rank)[0]


def slice_sizes(n_reads, slice_reads):
    """A rank's share of a job cut into batches of AT MOST `slice_reads` reads, all of the same size to within one read
    (12 500 reads in slices of 2000: seven slices of 1786 / 1785, not six of 2000 and one of 500 -- a short last batch runs at the
    rate of a small launch and a rank's wall time is the sum over its batches)."""
    n_reads, slice_reads = int(n_reads), int(slice_reads)
    if n_reads <= 0:
        return []
    if slice_reads <= 0:
        raise ValueError("slice_reads must be positive")
    k = (n_reads + slice_reads - 1) // slice_reads
    base, extra = divmod(n_reads, k)
    return [base + 1] * extra + [base] * (k - extra)


def merge_in_read_order(per_rank_indices, per_rank_results):
    """Inverse of shard_indices: results[i] for read i, whatever rank produced it."""
    n = sum(len(ix) for ix in per_rank_indices)
    out = [None] * n
    for ix, res in zip(per_rank_indices, per_rank_results):
        assert len(ix) == len(res)
        for i, r in zip(ix, res):
            assert out[int(i)] is None, "read %d aligned twice" % int(i)
            out[int(i)] = r
    assert all(o is not None for o in out), "a read was not aligned by any rank"
    return out
