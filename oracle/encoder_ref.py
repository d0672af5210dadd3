"""CPU oracle for the motion encoder (row N3) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement (torch CPU fp32) of the reference's ``Encoder.forward`` (stage1_VAE/modules/resnet3D.py:205-219)
with resnet18 BasicBlocks (:107-135, ``_make_layer`` :176-197).  Pinned against ``tests/golden/enc3d_*.npz``, which
``tests/golden/make_golden.py`` produced with the reference's own ``Encoder`` module."""
import torch
import torch.nn.functional as F


def _gn(sd, name, x):
    return F.group_norm(x, 16, sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def _basic_block(sd, p, x, stride_s, stride_t, has_down):
    """BasicBlock.forward -- resnet3D.py:120-135; downsample = Conv3d 3x3x3 with the block's strides + GroupNorm (:181-189)."""
    st = (stride_t, stride_s, stride_s)
    out = F.relu(_gn(sd, p + "bn1", F.conv3d(x, sd[p + "conv1.weight"], stride=st, padding=1)))
    out = _gn(sd, p + "bn2", F.conv3d(out, sd[p + "conv2.weight"], stride=1, padding=1))
    residual = x
    if has_down:
        residual = _gn(sd, p + "downsample.1", F.conv3d(x, sd[p + "downsample.0.weight"], stride=st, padding=1))
    return F.relu(out + residual)


def encoder(sd, x, channels=(64, 128, 256, 512, 512), stride_s=(1, 2, 2, 2), stride_t=(1, 2, 2, 2)):
    """Encoder.forward -- resnet3D.py:205-219 (use_max_pool False).  x [B,3,T,H,W] -> (mu [B,z], logvar [B,z])."""
    if x.size(1) > x.size(2):
        x = x.transpose(1, 2)
    x = F.relu(_gn(sd, "norm1", F.conv3d(x, sd["conv1.weight"], stride=(2, 2, 2), padding=(1, 3, 3))))
    inplanes = channels[0]
    for L, ch in enumerate(channels[1:]):
        x = _basic_block(sd, f"layer.{L}.0.", x, stride_s[L], stride_t[L], stride_s[L] != 1 or inplanes != ch)
        x = _basic_block(sd, f"layer.{L}.1.", x, 1, 1, False)
        inplanes = ch
    emb = x.squeeze(2)
    mu = F.conv2d(emb, sd["conv_mu.weight"], sd["conv_mu.bias"]).reshape(emb.size(0), -1)
    logvar = F.conv2d(emb, sd["conv_var.weight"], sd["conv_var.bias"]).reshape(emb.size(0), -1)
    return mu, logvar
