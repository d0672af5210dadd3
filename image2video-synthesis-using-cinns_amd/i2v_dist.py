"""Batch sharding of the sampling path over the GPUs of one node.

Every op on the path is per-sample, so the batch of start frames is split into contiguous shards, one per rank
(one process per GPU), weights replicated.  The latent draws are made for the GLOBAL batch and sliced, so results do
not depend on the GPU count.  The only collective is ONE all-gather of the per-rank ``[B/R, T, 3, H, W]`` blocks for
the final collation (RCCL over xGMI: ``torch.distributed`` backend "nccl"; the CPU tests use "gloo")."""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous shard [lo, hi) of ``total`` samples for ``rank``; the first ``total % world_size`` ranks take one extra."""
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(t, world_size, rank):
    lo, hi = shard_bounds(t.shape[0], world_size, rank)
    return t[lo:hi]


def collate(local, total, group=None):
    """All-gather the per-rank blocks into ``[total, ...]`` on every rank (rank order == sample order).
    Equal shards use one ``all_gather_into_tensor`` straight into the output buffer; ragged shards are padded to the
    largest shard and trimmed."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    ws = dist.get_world_size(group)
    if ws == 1:
        return local
    base, extra = divmod(total, ws)
    if extra == 0:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = base + 1
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((ws * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(total, ws, r)
        parts.append(buf[r * mx: r * mx + (hi - lo)])
    return torch.cat(parts, dim=0)


def synthesize_sharded(model_fn, x_0, residual, embed, group=None):
    """Runs ``model_fn(x_0_shard, residual_shard, embed_shard) -> [b, T, 3, H, W]`` on this rank's shard of the
    globally drawn inputs and collates the result on every rank."""
    if dist.is_available() and dist.is_initialized():
        ws, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        ws, rank = 1, 0
    total = x_0.shape[0]
    lo, hi = shard_bounds(total, ws, rank)
    local = model_fn(x_0[lo:hi], residual[lo:hi], embed[lo:hi])
    return collate(local, total, group)
