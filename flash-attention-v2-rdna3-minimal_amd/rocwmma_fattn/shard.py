"""Batch·head sharding of forward attention over the GPUs of one node (one process per GPU).

The reference is single-GPU (no torch.distributed / NCCL call anywhere; its launcher even allocates
L on the *current* device rather than q's, rocwmma_fattn/kernel_fp16.cu:795-796).  Forward attention
is independent per (batch, head) — the reference's grid is (B, H, Tr) with no inter-block
communication (kernel_fp16.cu:324-337) — so multi-GPU execution is a pure partition of the batch
dimension: rank r owns the contiguous slab [lo, hi) of B and runs the single-GPU operator on it.
No collective is on the data path.  RCCL (torch.distributed backend "nccl" on ROCm, xGMI links) is
used only at the edges, when a caller holds the full tensors on one rank:

    scatter_batch   root  -> every rank's slab      (dist.scatter; per-peer traffic 1/W of the tensor)
    gather_batch    slabs -> every rank / the root  (dist.all_gather_into_tensor / dist.gather)

Both work on any backend; the CPU tests drive them over gloo with world_size 2.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "local_batch", "scatter_batch", "gather_batch", "sharded_attention"]


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced partition of range(total): the first (total % world) ranks get one extra."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank %r/%r" % (world_size, rank))
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def local_batch(t, world_size, rank):
    """This rank's slab of a replicated [B, ...] tensor (a view, no copy)."""
    lo, hi = shard_bounds(t.size(0), world_size, rank)
    return t[lo:hi]


def _world(group):
    return dist.get_world_size(group), dist.get_rank(group)


def scatter_batch(t_full, shape, dtype, device, src=0, group=None, always_collective=False):
    """Root `src` holds t_full [B, ...]; every rank receives its contiguous batch slab.
    Non-root ranks pass t_full=None.  Slabs may be uneven (B % world != 0).  A group of one rank copies locally unless
    `always_collective` asks for the collective call anyway (bench.py --force-dist: runs RCCL on a 1-GPU box)."""
    world, rank = _world(group)
    B = shape[0]
    lo, hi = shard_bounds(B, world, rank)
    out = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=device)
    if world == 1 and not always_collective:
        out.copy_(t_full)
        return out
    if B % world == 0:
        chunks = list(t_full.contiguous().chunk(world, dim=0)) if rank == src else None
        dist.scatter(out, chunks, src=src, group=group)
        return out
    # uneven: one point-to-point transfer per peer (slabs differ in size, scatter needs equal sizes)
    if rank == src:
        reqs = []
        for r in range(world):
            rlo, rhi = shard_bounds(B, world, r)
            piece = t_full[rlo:rhi].contiguous()
            if r == src:
                out.copy_(piece)
            else:
                reqs.append(dist.isend(piece, dst=r, group=group))
        for q in reqs:
            q.wait()
    else:
        dist.recv(out, src=src, group=group)
    return out


def gather_batch(t_local, total_batch, group=None, always_collective=False):
    """Inverse of the partition: every rank ends with the full [B, ...] tensor."""
    world, rank = _world(group)
    if world == 1 and not always_collective:
        return t_local
    full = torch.empty((total_batch,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
    if total_batch % world == 0:
        dist.all_gather_into_tensor(full, t_local.contiguous(), group=group)
        return full
    pieces = []
    for r in range(world):
        rlo, rhi = shard_bounds(total_batch, world, r)
        pieces.append(full[rlo:rhi])
    # all_gather with uneven pieces: broadcast each slab from its owner
    for r in range(world):
        if r == rank:
            pieces[r].copy_(t_local)
        dist.broadcast(pieces[r], src=dist.get_global_rank(group, r) if group is not None else r, group=group)
    return full


def sharded_attention(q, k, v, causal=False, scale=None, group=None, attention_fn=None):
    """q, k, v: this rank's batch slab.  Runs the single-GPU operator on it; nothing is exchanged.
    `attention_fn(q, k, v, mask, causal, scale)` defaults to FlashAttentionFunction.apply — the hook
    exists so the partition logic can be exercised on CPU ranks in the gloo tests."""
    if attention_fn is None:
        from .FlashAttn import FlashAttentionFunction
        attention_fn = FlashAttentionFunction.apply
    return attention_fn(q, k, v, None, causal, scale)
