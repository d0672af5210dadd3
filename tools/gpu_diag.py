#!/usr/bin/env python
"""GPU bring-up diagnostics: compares every stage of the HIP path with the CPU oracle and prints rel-L2 errors
(no asserts -- one run localises a failing kernel).  Usage on the GPU box: python tools/gpu_diag.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "image2video-synthesis-using-cinns_amd"))
import i2v_native as native  # noqa: E402
import i2v_synth as synth  # noqa: E402
from oracle import decoder_ref, flow_ref  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def T(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def report(name, a, b):
    r = rel(a, b)
    print(f"  {name:40s} rel-L2 {r:.3e}  {'OK' if r < 1e-4 else 'BAD'}  (|ref| max {float(b.abs().max()):.3g}, nan {bool(torch.isnan(a).any())})")
    return r


def diag_flow():
    print("== flow")
    for emb, ctrl in ((64, False), (128, False), (94, True)):
        sd = T(synth.flow_state_dict(seed=7, embedding_dim=emb, control=ctrl))
        x = torch.randn(8, 64, generator=torch.Generator().manual_seed(1))
        e = torch.randn(8, emb, generator=torch.Generator().manual_seed(2))
        for nfl in (1, 20):
            sub = {k: v for k, v in sd.items() if int(k.split(".")[1]) < nfl}
            zt_ref, ld_ref = flow_ref.flow_forward(sub, x, e, n_flows=nfl, control=ctrl)
            z_ref = flow_ref.flow_reverse(sub, x, e, n_flows=nfl, control=ctrl)
            for graph in (False, True):
                h = native.NativeFlow(64, emb, 512, 2, nfl, control=1 if ctrl else 0, use_graph=graph)
                h.load(sub)
                zt, ld = h.forward(x.to(dev), e.to(dev))
                z = h.inverse(x.to(dev), e.to(dev))
                tag = f"E={emb} ctrl={int(ctrl)} n={nfl} graph={int(graph)}"
                report(tag + " fwd", zt, zt_ref.reshape(8, 64))
                report(tag + " logdet", ld, ld_ref)
                report(tag + " inv", z, z_ref.reshape(8, 64))
                rt = h.inverse(zt, e.to(dev))
                print(f"  {tag} round-trip max-abs {float((rt.cpu() - x).abs().max()):.3e}")


def oracle_blocks(sd, img, z, ups, upt):
    """Block-by-block oracle outputs (NCDHW) incl. intermediates of each block."""
    b = img.size(0)
    x = F.linear(z, sd["fc.weight"], sd["fc.bias"]).reshape(b, -1, 1, 4, 4)
    outs = []
    scales = [None, 2, 2, 2, (upt[0], ups[0], ups[0]), (upt[1], ups[1], ups[1])]
    for k, name in enumerate(("head_0", "g_0", "g_1", "g_2", "g_3", "g_4")):
        if scales[k] is not None:
            x = F.interpolate(x, scale_factor=scales[k])
        p = name + "."
        inter = {}
        a0 = F.leaky_relu(decoder_ref.spade(sd, p + "norm_0.", x, img, faithful=False), 0.2)
        inter[1] = a0
        dx = F.conv3d(a0, decoder_ref.sn_weight(sd, p + "conv_0"), sd[p + "conv_0.bias"], 1, 1)
        inter[2] = dx
        a1 = F.leaky_relu(decoder_ref.adain(sd, p + "norm_1.", dx, z), 0.2)
        inter[3] = a1
        x = decoder_ref.generator_block(sd, name, x, z, img, faithful=False)
        inter[5] = x
        outs.append(inter)
    return outs


def cl(x):  # NCDHW -> channels-last flat
    return x.permute(0, 2, 3, 4, 1).contiguous()


def diag_decoder(nf, ups, upt, img_size, B, seed=5, taps=True, mma=0, oracle=True):
    print(f"== decoder nf={nf} ups={ups} img={img_size} B={B} mma={mma}")
    sd = T(synth.decoder_state_dict(seed=seed, channel_factor=nf))
    img = 2 * torch.rand(B, 3, img_size, img_size, generator=torch.Generator().manual_seed(41)) - 1
    z = torch.randn(B, 64, generator=torch.Generator().manual_seed(42))
    h = native.NativeDecoder(nf, 64, ups, upt, True, mma=mma)
    h.load(sd)
    img_d, z_d = img.to(dev), z.to(dev)
    out = h.forward(img_d, z_d)
    torch.cuda.synchronize()
    if oracle:
        t0 = time.time()
        ref = decoder_ref.generator(sd, img, z, ups, upt, faithful=False)
        print(f"  oracle {time.time() - t0:.1f}s")
        report("final frames", out, ref)
    else:
        h0 = native.NativeDecoder(nf, 64, ups, upt, True, mma=0)
        h0.load(sd)
        report("final frames vs fp32-MFMA path", out, h0.forward(img_d, z_d))
        del h0
    if taps:
        folded = decoder_ref.fold_spectral_norm(sd)
        blocks = oracle_blocks(folded, img, z, ups, upt)
        for k in range(6):
            for which, nm in (((1, "lrelu(spade)"), (2, "conv_0"), (3, "lrelu(adain)"), (5, "block out")) if mma == 0 else ((2, "conv_0"), (5, "block out"))):
                r = cl(blocks[k][which])
                dst = torch.zeros(r.numel(), dtype=torch.float32, device=dev)
                h.debug_tap(k, which, dst)
                h.forward(img_d, z_d)
                torch.cuda.synchronize()
                report(f"block {k} {nm}", dst.view_as(r), r)
        h.debug_tap(0, 0, None)
    # timing
    for _ in range(2):
        h.forward(img_d, z_d)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 3
    for _ in range(n):
        h.forward(img_d, z_d)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    fl = h.flops_per_sample(img_size, img_size) * B
    print(f"  decoder time {dt * 1e3:.2f} ms  -> {B * 16 / dt:.1f} frames/s, {fl / dt / 1e12:.1f} TFLOP/s")


def time_flow():
    print("== flow timing")
    for B in (8, 64):
        sd = T(synth.flow_state_dict(seed=7, embedding_dim=64))
        for graph in (False, True):
            h = native.NativeFlow(64, 64, 512, 2, 20, use_graph=graph)
            h.load(sd)
            x = torch.randn(B, 64, device=dev)
            e = torch.randn(B, 64, device=dev)
            for _ in range(3):
                h.inverse(x, e)
            torch.cuda.synchronize()
            t0 = time.time()
            n = 20
            for _ in range(n):
                h.inverse(x, e)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            print(f"  B={B} graph={int(graph)} inverse {dt * 1e6:.1f} us  ({h.param_bytes / dt / 1e9:.1f} GB/s algorithmic)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["flow", "dec8", "dec8_128", "time_flow", "dec64"]
    print("device:", torch.cuda.get_device_name(0))
    if "flow" in which:
        diag_flow()
    if "time_flow" in which:
        time_flow()
    if "dec8" in which:
        diag_decoder(8, [2, 1], [2, 1], 64, 2)
    if "dec8_128" in which:
        diag_decoder(8, [2, 2], [2, 1], 128, 1, seed=6, taps=False)
    if "dec64" in which:
        diag_decoder(64, [2, 1], [2, 1], 64, 1, seed=7, taps=False)
    if "dec64b8" in which:
        diag_decoder(64, [2, 1], [2, 1], 64, 8, seed=7, taps=False)
    if "f16" in which:
        diag_decoder(8, [2, 1], [2, 1], 64, 2, mma=1)
        diag_decoder(8, [2, 2], [2, 1], 128, 1, seed=6, taps=False, mma=1)
        diag_decoder(64, [2, 1], [2, 1], 64, 1, seed=7, taps=False, mma=1)
        diag_decoder(32, [2, 2], [2, 1], 128, 1, seed=7, taps=False, mma=1)
        diag_decoder(64, [2, 1], [2, 1], 64, 16, seed=7, taps=False, mma=1, oracle=False)
        diag_decoder(64, [2, 1], [2, 1], 64, 16, seed=7, taps=False, mma=0, oracle=False)
