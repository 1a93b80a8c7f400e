"""Multi-GPU sharding of the clustering path.

(chrA,chrB) signal buckets are independent (tiddit_cluster.pyx:140-154 keeps no cross-bucket state),
so they are bin-packed onto ranks (one process per GPU) and clustered without any data-path
collective; ONE exchange step — a variable-count all-gather of the label arrays over RCCL/xGMI —
assembles the final cluster set on every rank.  The coverage / GC histograms shard by contig and
need no exchange (each rank returns its contigs' bins to the host caller).
"""
import numpy as np


def shard_buckets(sizes, world_size):
    """Longest-processing-time bin packing of buckets by signal count.
    -> list (per rank) of bucket indices, each ascending; deterministic on every rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    owned = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(o) for o in owned]


def shard_contigs(lengths, world_size):
    """Same packing for the histogram path (work ~ contig length)."""
    return shard_buckets(lengths, world_size)


def allgatherv(t, group=None):
    """Variable-count all-gather of 1-D tensors (RCCL has no allgatherv): counts are exchanged
    first, payloads are padded to the maximum and gathered with ONE all_gather_into_tensor.
    -> list of world_size tensors (rank order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    counts = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.cpu().tolist()
    mx = max(max(counts), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = torch.empty(world * mx, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx:r * mx + counts[r]] for r in range(world)]


def cluster_buckets_distributed(bucket_sizes, cluster_local, group=None, wire_dtype="int32"):
    """Cluster every bucket on its owner rank, then all-gather the labels.

    Labels are integers (-1 or a cluster id < 2^31) carried in float64 like the reference returns them; on
    the wire they travel as int32 (half the xGMI bytes, exact) unless wire_dtype is None.

    bucket_sizes : number of signals of every bucket (identical on all ranks)
    cluster_local(bucket_ids) -> 1-D float64 tensor with the labels of those buckets, concatenated
                   in the given order (on the GPU path: tdt_dbscan_device over the rank's buckets)
    -> list over ALL buckets of label tensors, identical on every rank.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owned = shard_buckets(bucket_sizes, world)
    import torch
    mine = cluster_local(owned[rank])
    out_dtype = mine.dtype
    if wire_dtype is not None:
        mine = mine.to(getattr(torch, wire_dtype))
    parts = [p.to(out_dtype) for p in allgatherv(mine, group)]
    labels = [None] * len(bucket_sizes)
    for r in range(world):
        off = 0
        for b in owned[r]:
            labels[b] = parts[r][off:off + int(bucket_sizes[b])]
            off += int(bucket_sizes[b])
    return labels
