"""Worker of tests/test_multi_rank.py: one rank of a world_size-2 gloo group.  Each rank plans (host side only)
the reads of its shard through the C ABI and reports the work; rank 0 checks the partition."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import signalalign_amd as sa  # noqa: E402
from signalalign_amd import shard, synth  # noqa: E402
import sa_cases as cases  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    alpha, k, t10, tab = synth.parse_model_table(cases.MODEL_6MER)
    n_reads = 12
    sizes = [300 + 170 * ((7 * i) % 11) for i in range(n_reads)]  # known to every rank
    mine = shard.shard_indices(sizes, rank, world)
    m = sa.Model.create(alpha, k, t10, tab)
    p = sa.default_params()
    res = []
    for i in mine:
        job = synth.make_read(int(i), sizes[int(i)], alpha, k, tab)
        info, regions, rows, segs = sa.plan_describe(m, p, job)
        res.append((int(i), float(info.cells_forward + info.cells_backward), int(info.n_segments)))
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine.tolist(), res))
    if rank == 0:
        merged = shard.merge_in_read_order([g[0] for g in gathered], [g[1] for g in gathered])
        assert [r[0] for r in merged] == list(range(n_reads))
        # the same plan, computed by one process
        for i in range(n_reads):
            job = synth.make_read(i, sizes[i], alpha, k, tab)
            info, _, _, _ = sa.plan_describe(m, p, job)
            assert merged[i][1] == info.cells_forward + info.cells_backward
        loads = [sum(sizes[i] for i in g[0]) for g in gathered]
        assert max(loads) <= 1.25 * min(loads), loads
        print("GLOO_OK", loads)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
