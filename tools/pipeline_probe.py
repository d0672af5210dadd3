"""Measurement (round 6): is the cINN pass of step k + 1 really hidden underneath the decoder of step k in the pipelined loop?
HIP events on the MAIN stream around every decoder forward of the bench loop (no profiler attached):
   idle_k    = end of decoder k - 1 (+ checksum)  ->  start of decoder k     (the main stream waits for the pass of step k here)
   decoder_k = start -> end of decoder k
and an event pair on the prefetch stream around every pass, placed on a common time axis through a shared origin event.  Prints per
configuration (B = 8, 64; prefetch priority high / normal) the medians and, per step, when the pass ran relative to the decoder it
should hide under."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "image2video-synthesis-using-cinns_amd")):
    sys.path.insert(0, p)
import i2v_pipeline  # noqa: E402
import i2v_synth as synth  # noqa: E402
from stage1_VAE.modules.decoder import Generator  # noqa: E402
from stage2_cINN.modules.flow_blocks import ConditionalFlow  # noqa: E402

torch.set_grad_enabled(False)
T = lambda sd: {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}  # noqa: E731
flow = ConditionalFlow(64, 64, 512, 2, 20, conditioning_option="None")
flow.load_state_dict(T(synth.flow_state_dict(seed=7, embedding_dim=64)))
gen = Generator({"channel_factor": 64, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
gen.load_state_dict(T(synth.decoder_state_dict(seed=7, channel_factor=64)))
flow, gen = flow.cuda().eval(), gen.cuda().eval()
E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
modes = sys.argv[1:] or ["-1", "0", "-1", "0"]
for prio in modes:
    os.environ["I2V_PREFETCH_PRIO"] = prio
    for B, n in ((8, 24), (64, 10)):
        x0, res, emb = synth.bench_inputs(B, 64, 64)
        x0, res, emb = x0.cuda(), res.cuda(), emb.cuda()
        pf = i2v_pipeline.LatentPrefetcher(lambda r, e: flow(r, e, reverse=True))
        ev = []

        def latent(r, e):                      # (runs under the prefetch stream)
            a = E(); a.record()
            z = flow(r, e, reverse=True)
            b = E(); b.record()
            ev[-1]["pass"] = (a, b)
            return z
        pf.latent_fn = latent
        origin = E()
        for rep in range(2):                   # rep 0 warms up
            ev.clear()
            torch.cuda.synchronize()
            origin.record()
            ev.append({})
            tk = pf.submit(res, emb)
            for k in range(n):
                z = pf.get(tk)
                cur = ev[-1]
                if k + 1 < n:
                    ev.append({})
                    tk = pf.submit(res, emb)
                s = E(); s.record()
                seq = gen.decode_sequence(x0, z.view(B, -1), 16)
                t = E(); t.record()
                chk = seq.view(torch.int64).sum()
                u = E(); u.record()
                cur["dec"] = (s, t, u)
            torch.cuda.synchronize()
        at = lambda e: origin.elapsed_time(e)  # noqa: E731
        idle, dec, step, hidden = [], [], [], []
        for k in range(2, n):
            s, t, u = ev[k]["dec"]
            idle.append(at(s) - at(ev[k - 1]["dec"][2]))
            dec.append(at(t) - at(s))
            step.append(at(u) - at(ev[k - 1]["dec"][2]))
            a, b = ev[k]["pass"]
            # the pass of step k should run inside decoder k - 1: [start of decoder k-1, end of decoder k-1]
            hidden.append((at(a) - at(ev[k - 1]["dec"][0]), at(b) - at(ev[k - 1]["dec"][0]), at(ev[k - 1]["dec"][1]) - at(ev[k - 1]["dec"][0])))
        med = lambda v: float(np.median(v))  # noqa: E731
        print(f"prefetch priority {prio:>2s}  B = {B:2d}: step {med(step):7.3f} ms = idle before the decoder {med(idle):6.3f} + decoder {med(dec):7.3f} (+ checksum); "
              f"pass of step k ran from {med([h[0] for h in hidden]):6.3f} to {med([h[1] for h in hidden]):6.3f} ms after the start of decoder k - 1 "
              f"(which took {med([h[2] for h in hidden]):6.3f} ms)", flush=True)
