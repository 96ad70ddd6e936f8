"""Multi-GPU sharding of a batch of independent pairs: contiguous pair ranges per rank, no data-path
collective; the only exchange is one all-gather of the int32 scores (RCCL on GPUs: backend "nccl" is RCCL on ROCm;
gloo on CPU for tests).  The reference has no multi-process code at all (pthreads in one address space)."""
import torch
import torch.distributed as dist


def shard_bounds(n_pairs, world, rank):
    """[lo, hi) of the contiguous shard of `rank`; shards differ by at most one pair."""
    q, r = divmod(int(n_pairs), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_scores(local, n_pairs, group=None):
    """All-gather the per-rank score shards (1-D int32 tensors of the shard_bounds sizes) into the full batch order.
    Equal shards take one all_gather_into_tensor; ragged shards are padded to the largest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_pairs, world, r)[1] - shard_bounds(n_pairs, world, r)[0] for r in range(world)]
    assert local.numel() == sizes[rank], (local.numel(), sizes[rank])
    mx = max(sizes)
    if min(sizes) == mx:
        out = torch.empty(mx * world, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    out = torch.empty(mx * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)])
