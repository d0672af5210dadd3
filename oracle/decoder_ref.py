"""CPU oracle for the stage-1 VAE decoder -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement (torch CPU fp32) of the reference's ``Generator`` and its blocks.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.

Parity pin: ``tests/golden/dec_*.npz`` generated from the reference's own modules by
``tests/golden/make_golden.py``; checked in ``tests/test_oracle_golden.py``.

``sd`` is a ``Generator.state_dict()``-shaped mapping {key: torch.Tensor}.  ``faithful=True``
re-derives W/sigma on every call and materialises the SPADE gamma/beta over T exactly like
the reference (the "faithful" CPU-baseline variant of SURVEY §8d); ``faithful=False`` uses
weights folded once by ``fold_spectral_norm`` (the "folded" variant).
"""
import torch
import torch.nn.functional as F


def sn_weight(sd, name):
    """torch.nn.utils.spectral_norm in eval mode (hook at decoder.py:20-25): no power
    iteration; weight = weight_orig / sigma, sigma = u . (W_mat v) -- signed, no abs (D6)."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    w = sd[name + ".weight_orig"]
    sigma = torch.dot(sd[name + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[name + ".weight_v"]))
    return w / sigma


def fold_spectral_norm(sd):
    """Return a copy of ``sd`` in which every spectral-normed conv carries a folded
    ``.weight`` instead of (weight_orig, u, v).  Done once at load time."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_orig"):
            name = k[: -len(".weight_orig")]
            out[name + ".weight"] = sn_weight(sd, name)
        elif k.endswith(".weight_u") or k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def _num_groups(c, g=16):
    while c % g != 0:  # normalization_layer.py:9-10
        g -= 1
    return g


def spade(sd, prefix, x, img, faithful=True):
    """Spade.forward -- stage1_VAE/modules/normalization_layer.py:18-24."""
    c = x.shape[1]
    normalized = F.group_norm(x, _num_groups(c), eps=1e-5)
    y = F.interpolate(img, mode="bilinear", size=x.shape[-2:], align_corners=True)
    y = F.leaky_relu(F.conv2d(y, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], 1, 1), 0.2)
    gamma = F.conv2d(y, sd[prefix + "conv_gamma.weight"], sd[prefix + "conv_gamma.bias"], 1, 1).unsqueeze(2)
    beta = F.conv2d(y, sd[prefix + "conv_beta.weight"], sd[prefix + "conv_beta.bias"], 1, 1).unsqueeze(2)
    if faithful:
        gamma = gamma.repeat_interleave(x.size(2), 2)
        beta = beta.repeat_interleave(x.size(2), 2)
    return normalized * (1 + gamma) + beta


def adain(sd, prefix, x, z):
    """ADAIN.forward -- normalization_layer.py:47-51 (gamma multiplies directly, no 1+)."""
    c = x.shape[1]
    out = F.instance_norm(x, eps=1e-5)
    gamma, beta = F.linear(z, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"]).chunk(2, 1)
    return gamma.view(-1, c, 1, 1, 1) * out + beta.view(-1, c, 1, 1, 1)


def norm3d(sd, prefix, x):
    """Norm3D.forward -- normalization_layer.py:33-35: GroupNorm(16, C, affine=True)."""
    return F.group_norm(x, 16, sd[prefix + "bn.weight"], sd[prefix + "bn.bias"], eps=1e-5)


def generator_block(sd, name, x, z, img, faithful=True):
    """GeneratorBlock.forward -- decoder.py:33-52."""
    p = name + "."
    learned = (p + "conv_s.weight_orig") in sd or (p + "conv_s.weight") in sd
    if learned:
        x_s = F.conv3d(norm3d(sd, p + "norm_s.", x), sn_weight(sd, p + "conv_s"))
    else:
        x_s = x
    dx = F.conv3d(F.leaky_relu(spade(sd, p + "norm_0.", x, img, faithful), 0.2),
                  sn_weight(sd, p + "conv_0"), sd[p + "conv_0.bias"], 1, 1)
    dx = F.conv3d(F.leaky_relu(adain(sd, p + "norm_1.", dx, z), 0.2),
                  sn_weight(sd, p + "conv_1"), sd[p + "conv_1.bias"], 1, 1)
    return x_s + dx


def generator(sd, img, motion, upsample_s=(2, 1), upsample_t=(2, 1), faithful=True, return_pre_tanh=False):
    """Generator.forward -- decoder.py:97-120.  img [B,3,H,W], motion [B,64] ->
    [B,16,3,H',W'] (contiguous here; the reference returns a transposed view)."""
    b = img.size(0)
    x = F.linear(motion, sd["fc.weight"], sd["fc.bias"]).reshape(b, -1, 1, 4, 4)
    x = generator_block(sd, "head_0", x, motion, img, faithful)
    for name in ("g_0", "g_1", "g_2"):
        x = F.interpolate(x, scale_factor=2)
        x = generator_block(sd, name, x, motion, img, faithful)
    x = F.interpolate(x, scale_factor=(upsample_t[0], upsample_s[0], upsample_s[0]))
    x = generator_block(sd, "g_3", x, motion, img, faithful)
    x = F.interpolate(x, scale_factor=(upsample_t[1], upsample_s[1], upsample_s[1]))
    x = generator_block(sd, "g_4", x, motion, img, faithful)
    pre = F.conv3d(F.leaky_relu(x, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=1)
    out = torch.tanh(pre).transpose(1, 2).contiguous()
    if return_pre_tanh:
        return out, pre
    return out
