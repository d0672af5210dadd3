"""Measurement (round 6): what do the launch gaps of a decoder forward cost at small batches?  The same forward eagerly (in-call overlap
on, the shipping path) and replayed from a torch.cuda.graph capture (everything inline on one stream, static input / output tensors;
the handles' ordering is capture-aware since round 6).  Prints ms per decoder pass for B = 8 and B = 64 (BAIR 64x64x16 nf = 64)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "image2video-synthesis-using-cinns_amd")):
    sys.path.insert(0, p)
import i2v_synth as synth  # noqa: E402
from stage1_VAE.modules.decoder import Generator  # noqa: E402

torch.set_grad_enabled(False)
dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.decoder_state_dict(seed=7, channel_factor=64).items()}
gen = Generator({"channel_factor": 64, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
gen.load_state_dict(dsd)
gen = gen.cuda().eval()
for B in (8, 64):
    x0, z, _ = synth.bench_inputs(B, 64, 64)
    x0, z = x0.cuda(), z.cuda()

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    n = 40 if B == 8 else 10
    eager = timed(lambda: gen(x0, z), n)
    os.environ["I2V_DEC_OVERLAP"] = "0"
    g0 = Generator({"channel_factor": 64, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
    g0.load_state_dict(dsd)
    g0 = g0.cuda().eval()
    os.environ.pop("I2V_DEC_OVERLAP")
    inline = timed(lambda: g0(x0, z), n)
    ref = gen(x0, z)
    xs, zs = x0.clone(), z.clone()
    gen(xs, zs)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ys = gen(xs, zs)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(ys, ref)
    replay = timed(graph.replay, n)
    print(f"B = {B}: decoder pass eager (in-call overlap) {eager:.3f} ms | eager inline {inline:.3f} ms | graph replay (inline) {replay:.3f} ms")
