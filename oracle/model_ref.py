"""CPU oracle for the inference facade -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates ``Model.forward`` (get_model.py:51-75) with the two inputs the reference draws or
computes internally made explicit so that "identical latent samples" is testable:
``residual`` (get_model.py:59, ``torch.randn`` on the CPU generator) and ``embed``
(INN.py:62, output of the ResNet-50 embedder, which is outside the hot path -- SURVEY §8f N1).

Parity: get_model.py / INN.py cannot be imported in the build container (omegaconf,
torchvision absent), so this restatement is pinned through its two constituents
(flow_ref / decoder_ref, both golden-pinned) and the fixture ``model_*.npz``, which
make_golden.py produces by driving the REFERENCE's ConditionalFlow and Generator modules
through the same get_model.py:65-75 sequence.
"""
import torch

from . import decoder_ref, flow_ref


def synthesize(flow_sd, dec_sd, x_0, residual, embed, vid_length=16, upsample_s=(2, 1), upsample_t=(2, 1),
               n_flows=20, control=False, faithful=True):
    """get_model.py:65-73 without the final batch slice: cINN inverse, decoder, autoregressive
    repeats on the last frame with the SAME z.  Returns [B, 16*ceil(L/16), 3, H, W]."""
    z = flow_ref.flow_reverse(flow_sd, residual, embed, n_flows=n_flows, control=control).view(x_0.size(0), -1)
    seq = decoder_ref.generator(dec_sd, x_0, z, upsample_s, upsample_t, faithful)
    while seq.shape[1] < vid_length:
        seq1 = decoder_ref.generator(dec_sd, seq[:, -1], z, upsample_s, upsample_t, faithful)
        seq = torch.cat((seq, seq1), dim=1)
    return seq


def model_forward(flow_sd, dec_sd, x_0, residual, embed, vid_length=16, **kw):
    """Model.forward -- get_model.py:51-75, including quirk Q3: the return slices dim 0
    (the BATCH), ``seq[:vid_length]``, not time."""
    return synthesize(flow_sd, dec_sd, x_0, residual, embed, vid_length, **kw)[:vid_length]
