"""Two-stage pipelining of the sampling path over a stream of batches (generate_samples.py:44-54 loops over batches of
start frames): the cINN inverse pass of batch k+1 -- an 82-launch dependent chain that leaves most of the chip idle --
runs on a high-priority side stream underneath the decoder pass(es) of batch k.

    pf = LatentPrefetcher(lambda res, emb: flow(res, emb, reverse=True))
    t = pf.submit(res_0, emb_0)
    for k in range(n):
        z = pf.get(t)
        if k + 1 < n: t = pf.submit(res_{k+1}, emb_{k+1})     # enqueued before the decoder of batch k
        frames_k = decoder(x0_k, z)

Every batch still gets exactly one cINN pass and one decoder run; only the order of enqueueing changes, and the latent
draws keep the order of the serial loop."""
import os

import torch


class LatentPrefetcher:
    def __init__(self, latent_fn, device=None, enabled=True):
        self.latent_fn = latent_fn
        self.enabled = bool(enabled) and torch.cuda.is_available()
        # high priority: the chain's 82 small dependent launches get their workgroups dispatched in front of the decoder's big grids
        # (I2V_PREFETCH_PRIO=0: same priority as the caller's stream, for A/B runs)
        prio = int(os.environ.get("I2V_PREFETCH_PRIO", "-1"))
        self.stream = torch.cuda.Stream(device=device, priority=prio) if self.enabled else None

    def mark(self):
        """An event on the current stream: "everything enqueued so far is complete".  Pass it to ``submit(..., _ready=ev)`` when the
        arguments of the next pass are complete NOW but the submit itself comes later (behind the decoder's launches, see the
        shared-side-stream order below) -- the pass then does not wait for those launches."""
        if not self.enabled:
            return None
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def submit(self, *args, _ready=None, **kwargs):
        """Enqueue ``latent_fn(*args)`` on the side stream (after everything already enqueued on the current stream, so
        the arguments are complete; or after the event ``_ready`` of an earlier ``mark()``).  Returns a ticket for ``get``.

        When the decoder shares this stream for its own side work (``Generator.share_side_stream(pf.stream)``: ONE side stream per
        job, what a rank of a multi-GPU job should run) the usual order -- submit batch k + 1, then the decoder of batch k -- stays:
        the decoder handle then runs its two tiny first SPADE levels inline, so the main chain does not wait for the pass queued in
        front of its side work.  ``_ready`` lets a caller enqueue the pass BEHIND the decoder's launches without making it wait for
        them (``ev = pf.mark(); frames = decoder(...); t = pf.submit(..., _ready=ev)``)."""
        if not self.enabled:
            return (self.latent_fn(*args, **kwargs), None)
        ready = _ready
        if ready is None:
            ready = torch.cuda.Event()
            ready.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            z = self.latent_fn(*args, **kwargs)
            done = torch.cuda.Event()
            done.record()
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a) and a.is_cuda:
                a.record_stream(self.stream)      # the caller may drop its reference while the side stream still reads it
        return (z, done)

    def get(self, ticket):
        """The latent of a ticket, usable on the current stream."""
        z, done = ticket
        if done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            if torch.is_tensor(z):
                z.record_stream(cur)
        return z
