"""world_size-2 gloo tests of the multi-GPU sharding path (CPU; the per-rank compute is the oracle,
standing in for the HIP kernels that the same code drives on the GPU box)."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_buckets_lpt():
    from tiddit_amd.dist import shard_buckets
    sizes = [100, 1, 50, 50, 7, 0, 30]
    sh = shard_buckets(sizes, 3)
    assert sorted(sum(sh, [])) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in s) for s in sh]
    assert loads == [100, 80, 58]
    assert shard_buckets(sizes, 1) == [list(range(len(sizes)))]
    assert shard_buckets([], 4) == [[], [], [], []]


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    import oracle
    from tiddit_amd.dist import allgatherv, cluster_buckets_distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parts = allgatherv(torch.arange(rank * 3 + 1, dtype=torch.float64) + 10 * rank)
        assert [p.numel() for p in parts] == [r * 3 + 1 for r in range(world)]
        assert all(torch.equal(parts[r], torch.arange(r * 3 + 1, dtype=torch.float64) + 10 * r) for r in range(world))
        empty = allgatherv(torch.zeros(0, dtype=torch.float64))
        assert all(p.numel() == 0 for p in empty)
        rng = np.random.default_rng(5)           # same buckets on every rank
        sizes = [0, 40, 700, 3, 1200, 90, 1, 350]
        buckets = []
        for s in sizes:
            x = np.sort(rng.integers(0, max(10, s * 30), s))
            y = x + rng.integers(0, 900, s)
            buckets.append(np.stack([x, y], 1).astype(np.int64).reshape(s, 2))

        def cluster_local(ids):
            out = [oracle.dbscan_main(buckets[b], 300, 3) if sizes[b] else np.zeros(0) for b in ids]
            return torch.from_numpy(np.concatenate(out) if out else np.zeros(0))

        labels = cluster_buckets_distributed(sizes, cluster_local)
        for b, s in enumerate(sizes):
            want = oracle.dbscan_main(buckets[b], 300, 3) if s else np.zeros(0)
            assert np.array_equal(labels[b].numpy(), want), b
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_allgatherv_and_bucket_sharding_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
