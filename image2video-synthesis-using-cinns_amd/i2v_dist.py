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


def collate(local, total, group=None, out=None):
    """All-gather the per-rank blocks into ``[total, ...]`` on every rank (rank order == sample order).
    Equal shards use one ``all_gather_into_tensor`` straight into the output buffer (``out``: optional pre-allocated
    ``[total, ...]`` buffer); ragged shards -- including EMPTY ones when total < world size -- are padded to the largest
    shard and trimmed."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    ws = dist.get_world_size(group)
    if ws == 1:
        return local
    base, extra = divmod(total, ws)
    shape = tuple(local.shape[1:])
    if extra == 0:
        if out is None:
            out = torch.empty((total,) + shape, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = base + 1
    pad = torch.zeros((mx,) + shape, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((ws * mx,) + shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(total, ws, r)
        parts.append(buf[r * mx: r * mx + (hi - lo)])
    return torch.cat(parts, dim=0)


class OverlappedCollator:
    """The final sequence collation of a steady stream of steps, taken off the compute stream.

    ``submit(local)`` records an event on the current (compute) stream, makes a side stream wait for it and issues the
    RCCL all-gather of the rank's ``[B/R, T, 3, H, W]`` block there, into one of two pre-allocated ``[B, T, 3, H, W]``
    buffers (allocated once, alternating); the compute stream goes straight on with the next step's cINN pass and
    decoder, which overlap the transfer over xGMI.  ``result()`` makes the current stream wait for the newest gather and
    returns its buffer; a buffer is reused two submits later, by which time the caller has consumed it."""

    def __init__(self, total, group=None, emulate=False, stream=None):
        self.total, self.group = total, group
        # stream (optional): issue the gathers on THIS side stream instead of one of the collator's own -- e.g. the stream the cINN
        # prefetch of the NEXT step runs on (i2v_pipeline.LatentPrefetcher.stream): a rank then runs main + the decoder handle's side
        # stream + ONE more stream, the count that fits HIP's four hardware queues.  The gather of step k and the pass
        # of step k + 2 serialise on it with a whole step of slack.
        self._ext_stream = stream
        # emulate (one process, measurement only): the same side stream, events and double buffers, with a device copy of the block
        # standing in for the RCCL all-gather -- the stream configuration of an N > 1 job (main + side streams + the collation stream)
        # on ONE GPU, to measure what the extra stream costs on HIP's four hardware queues (bench.py --emulate-collation).
        # emulate = "rccl": the REAL path below on a one-rank RCCL process group (torch.distributed / RCCL issue the gather themselves)
        self.emulate = emulate or False
        self.stream = None
        self.bufs = [None, None]
        self.events = [None, None]
        self.i = 0
        self.last = None

    def submit(self, local):
        if self.emulate and local.is_cuda and not (dist.is_available() and dist.is_initialized()):
            return self._submit_emulated(local)
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.group) == 1 and self.emulate != "rccl") \
                or self.total % dist.get_world_size(self.group) != 0 or not local.is_cuda:
            self.last = ("sync", collate(local, self.total, self.group))  # ragged shards / CPU tensors (gloo): plain path
            return
        if self.stream is None:
            self.stream = self._ext_stream if self._ext_stream is not None else torch.cuda.Stream(device=local.device)
        k = self.i
        self.i ^= 1
        if self.bufs[k] is None or self.bufs[k].shape[1:] != local.shape[1:] or self.bufs[k].dtype != local.dtype:
            if self.events[k] is not None:
                self.events[k].synchronize()             # a gather into the old buffer may still be in flight
            with torch.cuda.stream(self.stream):         # owned by the side stream (the only writer); readers sync via events
                self.bufs[k] = torch.empty((self.total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = local.contiguous()
        ready = torch.cuda.Event()
        ready.record()                                   # the rank's block is complete at this point of the compute stream
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            dist.all_gather_into_tensor(self.bufs[k], local, group=self.group)
            local.record_stream(self.stream)             # keep the block alive until the gather has read it
            done = torch.cuda.Event()
            done.record()
        self.events[k] = done
        self.last = ("async", k)

    def _submit_emulated(self, local):
        if self.stream is None:
            self.stream = self._ext_stream if self._ext_stream is not None else torch.cuda.Stream(device=local.device)
        k = self.i
        self.i ^= 1
        if self.bufs[k] is None or self.bufs[k].shape != local.shape:
            if self.events[k] is not None:
                self.events[k].synchronize()
            with torch.cuda.stream(self.stream):
                self.bufs[k] = torch.empty_like(local)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.bufs[k].copy_(local)                    # stands in for the all-gather kernel
            local.record_stream(self.stream)
            done = torch.cuda.Event()
            done.record()
        self.events[k] = done
        self.last = ("async", k)

    def result(self):
        if self.last is None:
            raise RuntimeError("OverlappedCollator.result() before submit()")
        kind, v = self.last
        if kind == "sync":
            return v
        cur = torch.cuda.current_stream()
        cur.wait_event(self.events[v])
        self.bufs[v].record_stream(cur)                  # read on the caller's stream, allocated on the side stream
        return self.bufs[v]


def synthesize_sharded(model_fn, x_0, residual, embed, group=None):
    """Runs ``model_fn(x_0_shard, residual_shard, embed_shard) -> [b, T, 3, H, W]`` on this rank's shard of the
    globally drawn inputs and collates the result on every rank."""
    if dist.is_available() and dist.is_initialized():
        ws, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        ws, rank = 1, 0
    total = x_0.shape[0]
    lo, hi = shard_bounds(total, ws, rank)
    emb = embed[lo:hi] if embed is not None else None
    if hi > lo:
        local = model_fn(x_0[lo:hi], residual[lo:hi], emb)
        shape = torch.tensor(list(local.shape[1:]), dtype=torch.int64, device=local.device)
    else:
        local, shape = None, None
    if ws > 1 and total < ws:
        # more ranks than samples: the ranks with an empty shard skip the model (the native code rejects batch 0) and
        # contribute a zero-row block; they learn the block shape from rank 0, which always holds a sample
        dev = x_0.device if local is None else local.device
        if shape is None:
            shape = torch.zeros(4, dtype=torch.int64, device=dev)
        dist.broadcast(shape, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if local is None:
            local = torch.zeros((0,) + tuple(int(v) for v in shape.tolist()), dtype=x_0.dtype, device=dev)
    elif local is None:
        raise ValueError("synthesize_sharded: empty batch")
    return collate(local, total, group)
