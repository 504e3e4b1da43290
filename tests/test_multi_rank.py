"""N > 1 path on CPU: world_size 2, gloo.  Reads shard across ranks with no data-path collective; the only
communication is gathering results (here: plan statistics, the GPU compute being unavailable on CPU)."""
import os
import socket
import subprocess
import sys

import numpy as np

from signalalign_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_is_a_partition_and_balanced():
    rng = np.random.default_rng(0)
    costs = rng.integers(500, 20000, size=1000)
    for world in (1, 2, 4, 8):
        parts = [shard.shard_indices(costs, r, world) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(1000))
        loads = [costs[p].sum() for p in parts]
        assert max(loads) <= 1.02 * min(loads)
    res = shard.merge_in_read_order(parts, [[int(i) * 2 for i in p] for p in parts])
    assert res == [2 * i for i in range(1000)]


def test_the_scaling_job_is_dealt_completely_and_evenly():
    # BASELINE configs[4] as bench.py's config.scaling_job deals it: 100 000 reads of 10 000 events over 1, 2, 4, 8 ranks, each
    # rank's share cut into slices of 2000 reads (bench.py --job-slice)
    total, ev, sl = 100000, 10000, 2000
    for world in (1, 2, 4, 8):
        parts = [shard.shard_indices([ev] * total, r, world) for r in range(world)]
        assert sum(len(p) for p in parts) == total and len(set(len(p) for p in parts)) == 1
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(total))
        sizes = shard.slice_sizes(len(parts[0]), sl)
        assert sum(sizes) == total // world and max(sizes) <= sl and max(sizes) - min(sizes) <= 1
        assert len(sizes) == -(-(total // world) // sl)   # no more batches than slices of `sl` reads would be
    # equal slices (round 6): a rank's last batch is as large as its others
    assert shard.slice_sizes(12500, 2000) == [1786] * 5 + [1785] * 2 and shard.slice_sizes(0, 5) == [] and shard.slice_sizes(3, 5) == [3]
    assert shard.slice_sizes(100000, 2000) == [2000] * 50 and shard.slice_sizes(7, 2) == [2, 2, 2, 1]


def test_two_rank_gloo_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert pr.returncode == 0, pr.stdout + pr.stderr
    assert "GLOO_OK" in pr.stdout


def test_manifest_split_for_the_batch_runner(tmp_path):
    # the CLI front door shards a manifest the same way the library shards reads: every read exactly once, balanced
    from signalalign_amd import batch_runner
    lines = []
    for i in range(37):
        p = tmp_path / ("r%d.npRead" % i)
        p.write_bytes(b"x" * (1000 + 137 * ((i * 7) % 11)))
        lines.append("\t".join(["read%d" % i, str(p), "g.cigar", "out%d.tsv" % i]))
    lines.append("\t".join(["gone", str(tmp_path / "missing.npRead"), "g.cigar", "x.tsv"]))
    mf = tmp_path / "m.tsv"
    mf.write_text("# header comment\n" + "\n".join(lines) + "\n\n")
    got = batch_runner.read_manifest(str(mf))
    assert got == lines
    parts = batch_runner.split_manifest(got, 8)
    assert sorted(sum(parts, [])) == sorted(lines) and len(parts) == 8
    sizes = [sum(os.path.getsize(ln.split("\t")[1]) for ln in part if os.path.exists(ln.split("\t")[1])) for part in parts]
    assert max(sizes) - min(sizes) <= 2500
