"""Clip-level data parallelism: one process per GPU, full weight replica each, no collective on the data path.

The reference is single-process; every clip / 30 s chunk / language replica is
independent (§8e), so the build deals work items to ranks by cost and, at the end of a batch, moves the per-clip tag
tensors (~24 KB per 30 s clip) to rank 0 with ONE collective: `ids | maxprob | offsets` are packed bit-for-bit
into a single int32 [n, T, 4] payload and gathered over RCCL (backend "nccl" on ROCm; gloo in the CPU tests).
The product loop does not even pack: `TagBatch.packed` (tagger.py) already is one contiguous int32 blob
`ids [n*T] | maxprob bits [n*T] | offsets bits [n*T*2] | status word`, which `gather_packed` moves as it stands.
"""
from __future__ import annotations

import torch


def _dist():
    import torch.distributed as dist
    return dist


def shard_items(costs, world: int):
    """Longest-processing-time-first dealing of work items to `world` ranks.
    costs: per-item cost (e.g. samples).  Returns a list (per rank) of item indices, each in ascending order.
    Deterministic: ties break on the item index, so every rank computes the same plan without communicating."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    plan = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        plan[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in plan]


def pack_tags(ids: torch.Tensor, maxprob: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """[n,T] int32, [n,T] f32, [n,T,2] f32 -> one int32 [n,T,4] payload (bit-exact views)."""
    n, T = ids.shape
    out = torch.empty(n, T, 4, dtype=torch.int32, device=ids.device)
    out[..., 0] = ids.to(torch.int32)
    out[..., 1] = maxprob.contiguous().view(torch.int32)
    out[..., 2:] = offsets.contiguous().view(torch.int32)
    return out


def unpack_tags(payload: torch.Tensor):
    ids = payload[..., 0].contiguous()
    maxprob = payload[..., 1].contiguous().view(torch.float32)
    offsets = payload[..., 2:].contiguous().view(torch.float32)
    return ids, maxprob, offsets


def gather_tags(ids, maxprob, offsets, dst: int = 0, counts=None, group=None):
    """Gather every rank's tag tensors on `dst` (rank order).  Returns (ids, maxprob, offsets) concatenated over
    ranks on `dst`, and the local tensors unchanged elsewhere.  `counts` (items per rank) allows unequal shards:
    payloads are padded to max(counts) for the collective and trimmed on arrival."""
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids, maxprob, offsets
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    payload = pack_tags(ids, maxprob, offsets)
    n, T = ids.shape
    if counts is not None:
        nmax = max(counts)
        if n != counts[rank]:
            raise ValueError("counts[rank] does not match the local shard")
        if n < nmax:
            pad = torch.zeros(nmax - n, T, 4, dtype=torch.int32, device=payload.device)
            payload = torch.cat([payload, pad])
    else:
        counts = [n] * world
        nmax = n
    bufs = [torch.empty(nmax, T, 4, dtype=torch.int32, device=payload.device) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return ids, maxprob, offsets
    full = torch.cat([b[:c] for b, c in zip(bufs, counts)])
    return unpack_tags(full)


def split_packed(blob: torch.Tensor, n: int, T: int):
    """Views into one `TagBatch.packed` blob (4*n*T + 1 int32 words) -> (ids [n,T] i32, maxprob [n,T] f32,
    offsets [n,T,2] f32, status word [1] i32).  No copies."""
    nt = n * T
    if blob.numel() != 4 * nt + 1:
        raise ValueError("blob does not hold n x T packed tags")
    return (blob[0:nt].view(n, T), blob[nt:2 * nt].view(torch.float32).view(n, T),
            blob[2 * nt:4 * nt].view(torch.float32).view(n, T, 2), blob[4 * nt:4 * nt + 1])


def gather_packed(blob: torch.Tensor, dst: int = 0, out=None, group=None):
    """ONE collective for a step's tags: every rank's contiguous `TagBatch.packed` blob (equal sizes: ranks label equal
    batches; a short last batch is padded by its owner) lands on `dst` as rows of `out` ([world, words] int32, allocated
    here when None).  Returns `out` on `dst` (row r = rank r's blob, split with `split_packed`), None elsewhere; with one
    rank, the blob itself as a [1, words] view."""
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return blob.view(1, -1)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank != dst:
        dist.gather(blob, None, dst=dst, group=group)
        return None
    if out is None:
        out = torch.empty(world, blob.numel(), dtype=blob.dtype, device=blob.device)
    dist.gather(blob, [out[r] for r in range(world)], dst=dst, group=group)
    return out
