"""The product's RCCL code path (csrc/tdt_comm.hip behind tiddit_amd/comm.py: tdt_comm_init, tdt_allgatherv with its group of
broadcasts, tdt_allreduce_sum_f64) with N > 1 ranks.  The GPU boxes have ONE device and RCCL refuses two ranks on one device, so the
ranks bind tests/rccl_standin/ (TIDDIT_RCCL_LIB) — the nccl* entry points over shared memory + hipMemcpy — and share GPU 0.  What runs is
the product's own plan (tdt_allgatherv_plan), displacement arithmetic, call sequence and stream handling; what is replaced is the wire."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("rccl") / "librccl_standin.so")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out,
                           os.path.join(REPO, "tests", "rccl_standin", "rccl_standin.cpp"), "-lrt"])
    return out


def _buckets(seed=4):
    rng = np.random.default_rng(seed)
    sizes = [0, 700, 40, 9000, 3, 1200, 90, 1, 350, 20000, 5, 64, 65, 8000]
    out = []
    for s in sizes:
        x = np.sort(rng.integers(0, max(10, s * 40), s))
        y = x + rng.integers(0, 900, s)
        out.append(np.stack([x, y], 1).astype(np.int64).reshape(s, 2))
    return out


def _rank(rank, world, port, q, lib, uid_q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), TIDDIT_HIP_DEVICE="0", TIDDIT_RCCL_LIB=lib)
    try:
        import torch
        import torch.distributed as dist
        from tiddit_amd import _native, comm, dist as tdist, tiddit_cluster
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        ctx = _native.default_context()
        c = comm.Comm.from_torch(ctx)                                   # the 128-byte id travels over the gloo group, the data over "RCCL"
        buckets = _buckets()
        sizes = [len(b) for b in buckets]
        owned = tdist.shard_buckets(sizes, world)
        labs = tiddit_cluster.cluster_buckets([buckets[i] for i in owned[rank]], 300, 3, ctx=ctx)
        mine = np.concatenate(labs).astype(np.int32) if labs else np.zeros(0, dtype=np.int32)
        counts = [sum(sizes[i] for i in owned[r]) for r in range(world)]
        d_send = torch.from_numpy(mine).to(dev)
        d_recv = torch.full((sum(counts) + 7,), -77, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.allgatherv(d_send.data_ptr(), d_recv.data_ptr(), counts, 4)
        ctx.sync()
        got = d_recv.cpu().numpy()
        assert (got[sum(counts):] == -77).all()                         # nothing written behind the last rank's range
        parts = np.split(got[:sum(counts)], np.cumsum(counts)[:-1])
        by_bucket = tdist.split_gathered(sizes, owned, parts)
        # ... the same exchange through torch.distributed (gloo) and the single-GPU labels
        gl = tdist.allgatherv(torch.from_numpy(mine))
        whole = tiddit_cluster.cluster_buckets(buckets, 300, 3, ctx=ctx)
        for b in range(len(buckets)):
            assert np.array_equal(by_bucket[b], whole[b].astype(np.int32)), b
        assert all(np.array_equal(parts[r], gl[r].numpy()) for r in range(world))
        # a rank with nothing to send, and element sizes other than 4
        counts2 = [0 if r == 1 else 1000 + r for r in range(world)]
        s2 = torch.arange(counts2[rank], dtype=torch.float64, device=dev) + 1000 * rank
        r2 = torch.zeros(sum(counts2), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        c.allgatherv(s2.data_ptr() if counts2[rank] else None, r2.data_ptr(), counts2, 8)
        ctx.sync()
        want2 = np.concatenate([np.arange(counts2[r], dtype=np.float64) + 1000 * r for r in range(world)])
        assert np.array_equal(r2.cpu().numpy(), want2)
        # the exact all-reduce of coverage bins (multiples of 2^-S below 2^53: any order gives the same float64)
        rng = np.random.default_rng(100 + rank)
        n = 1_300_001
        bins = rng.integers(0, 1 << 30, n).astype(np.float64) / 1024.0
        d = torch.from_numpy(bins).to(dev)
        torch.cuda.synchronize()
        c.allreduce_sum_f64(d.data_ptr(), n)
        ctx.sync()
        want = sum((np.random.default_rng(100 + r).integers(0, 1 << 30, n).astype(np.float64) / 1024.0) for r in range(world))
        assert np.array_equal(d.cpu().numpy(), want)
        c.close()
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except BaseException:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_tdt_comm_with_n_ranks_on_one_gpu(standin, world):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, standin, None)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(v == "ok" for v in res.values()), res


_RCCL_ONE_RANK = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np
import torch
import torch.distributed as dist
from tiddit_amd import dist as tdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and tdist._wire_device().type == "cuda"
rng = np.random.default_rng(3)
# alltoall_bytes (all_to_all_single on device tensors): empty, small, and a payload cut into many messages
for cap, n in ((None, 0), (None, 1), (None, 3_000_001), (4096, 1_000_003), (1, 37)):
    if cap is None:
        os.environ.pop("TIDDIT_WIRE_MAX_BYTES", None)
    else:
        os.environ["TIDDIT_WIRE_MAX_BYTES"] = str(cap)
    part = rng.integers(0, 256, n, dtype=np.uint8)
    got = tdist.alltoall_bytes([part])
    assert len(got) == 1 and np.array_equal(np.asarray(got[0]), part), (cap, n)
    blobs = tdist.gather_bytes(part.tobytes(), dst=0)
    assert blobs == [part.tobytes()]
    obj = tdist.broadcast_object({"n": n, "part": part.tolist()[:1000], "nested": [None, 1.5, "x"]}, src=0)
    assert obj["n"] == n and obj["nested"] == [None, 1.5, "x"]
os.environ.pop("TIDDIT_WIRE_MAX_BYTES", None)
assert tdist.allgather_i64([7, -1, 1 << 40]).tolist() == [[7, -1, 1 << 40]]
# the seam table: one shard, a shard that starts at the header, an empty one
assert tdist.check_seams(0, 0, False) == [[0, 0, 0]]
assert tdist.check_seams(None, None, True) == [[-1, -1, 1]]
# the all-reduce of the bins on a device tensor: exact (a sum over one rank), in place, 60 M bins = the 50-bp genome
bins = torch.arange(60_000_000, dtype=torch.float64, device="cuda") * 0.25
want = bins.clone()
out = tdist.allreduce_bins(bins)
assert out.data_ptr() == bins.data_ptr() and torch.equal(bins, want)
# the variable-count all-gather of the cluster set
t = torch.arange(12345, dtype=torch.int64, device="cuda")
parts = tdist.allgatherv(t)
assert len(parts) == 1 and torch.equal(parts[0], t)
assert len(tdist.allgatherv(t[:0])) == 1
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
'''


def test_dist_helpers_over_real_rccl_with_one_rank():
    """The byte exchanges of the N-rank job — alltoall_bytes (one all_to_all_single per TIDDIT_WIRE_MAX_BYTES of the largest piece),
    gather_bytes, broadcast_object, allgather_i64, check_seams, allreduce_bins, allgatherv — called DIRECTLY over backend nccl = RCCL
    with device tensors (one rank: RCCL refuses two ranks on one device), incl. empty payloads and payloads cut into many messages.
    The same functions run with 2 and 3 ranks over gloo in tests/test_dist_cpu.py; only the whole `--sv` job reached them over RCCL before."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % {"repo": repo}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stderr[-3000:]
