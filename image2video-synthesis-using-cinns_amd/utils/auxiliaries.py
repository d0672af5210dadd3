"""Output side of the sampling path: the on-disk tiling of generated sequences and the prior-sampling
loop that the reference's evaluation runs over a data loader.

Own implementations of the reference behaviour (``utils/auxiliaries.py:15-22`` GIF tiling, ``:53-55`` denormalisation,
``:87-101`` the sampling loop of ``evaluate_FVD_prior``); the FVD networks, wandb logging and video writers that follow
that loop in the reference are outside the hot path and are not rebuilt."""
import numpy as np
import torch


def denorm(x):
    """[-1, 1] -> [0, 1], clipped (reference semantics; the input is left untouched)."""
    return torch.clamp(x * 0.5 + 0.5, 0.0, 1.0)


def convert_seq2gif(sequence):
    """``[N, T, 3, H, W]`` in [-1, 1] -> float array ``[T, H, N*W, 3]``: the N clips side by side along the width, scaled
    so that the brightest value of the whole strip is 255 (the reference divides by the strip's own maximum, not by 1)."""
    n, t, c, h, w = sequence.shape
    strip = denorm(sequence.detach().float().cpu())          # [N, T, C, H, W]
    strip = strip.permute(1, 3, 0, 4, 2).reshape(t, h, n * w, c).numpy()
    peak = float(strip.max())
    return strip * (255.0 / peak)


@torch.no_grad()
def sample_prior(dloader, cINN, decoder, z_dim, control=False, generator=None):
    """The sampling loop of the reference's ``evaluate_FVD_prior`` (second caller of cINN^-1 + decoder): for every batch
    ``file`` of ``dloader`` (a dict with ``"seq"`` ``[B, T+1, 3, H, W]`` and, with ``control``, ``"cond"`` ``[B, 3]``) draw
    ``res ~ N(0, 1)`` on the CPU generator, invert the cINN conditioned on the first frame, decode, and collect
    ``(generated [N, 16, 3, H, W], original seq[:, 1:])`` on the CPU.  ``cINN`` is a ``SupervisedTransformer``-like callable
    ``cINN(res, cond, reverse=True)``, ``decoder`` a ``Generator``-like callable ``decoder(x_0, z)``."""
    gen, orig = [], []
    for file in dloader:
        seq = file["seq"].float().cuda()
        b = seq.size(0)
        res = torch.randn(b, z_dim, generator=generator).cuda()
        x_0 = seq[:, 0].contiguous()
        cond = [x_0, file["cond"]] if control else [x_0]
        z = cINN(res, cond, reverse=True).view(b, -1)
        gen.append(decoder(x_0, z).cpu())
        if hasattr(decoder, "native") and decoder.native().status():   # (.cpu() synchronised) range guard of the split-fp16 operands
            raise RuntimeError("sample_prior: the decoder's activations left the fp16 range of the split-fp16 conv operands "
                               "(use mma = 0 for this checkpoint)")
        orig.append(seq[:, 1:].cpu())
    return torch.cat(gen, dim=0), torch.cat(orig, dim=0)
