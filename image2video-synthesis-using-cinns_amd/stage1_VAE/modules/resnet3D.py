"""Mirror of the motion encoder of the reference's ``stage1_VAE/modules/resnet3D.py`` (``Encoder``, lines 138-219; row N3
of the coverage contract): 3D ResNet-18 with GroupNorm(16), ``conv_mu`` / ``conv_var`` heads.  The modules carry the
reference's state_dict keys (``conv1.weight``, ``norm1.*``, ``layer.{L}.{i}.conv{1,2}.weight``, ``.bn{1,2}.*``,
``.downsample.{0.weight,1.*}``, ``conv_mu.*``, ``conv_var.*``); the arithmetic runs in libi2v_hip.so (csrc/i2v_encoder.hip).
The discriminators of the same reference file are training-only and out of scope."""
import torch
import torch.nn as nn

import i2v_native as native
from i2v_params import AffineParams, ConvParams, NativeBacked, _NoForward


class BasicBlock(_NoForward):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, stride_t=1, downsample=None, spectral=False):
        super().__init__()
        if spectral:
            raise NotImplementedError("the Encoder never enables spectral norm (resnet3D.py:153)")
        self.conv1 = ConvParams(inplanes, planes, 3, 3, bias=False)
        self.bn1 = AffineParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, 3, bias=False)
        self.bn2 = AffineParams(planes)
        if downsample is not None:
            self.downsample = downsample
        self.stride = stride


class _StemConv(_NoForward):
    def __init__(self, cout):
        super().__init__()
        w = torch.empty(cout, 3, 3, 7, 7)
        nn.init.kaiming_normal_(w, mode="fan_out")
        self.weight = nn.Parameter(w)


class Encoder(NativeBacked):
    def __init__(self, dic):
        super().__init__()
        if dic["res_type_encoder"] != "resnet18":
            raise NotImplementedError("Encoder: every shipped config uses res_type_encoder 'resnet18'")
        self.use_max_pool = dic["use_max_pool"]
        if self.use_max_pool:
            raise NotImplementedError("Encoder: use_max_pool is false in every shipped config")
        self.z_dim = dic["z_dim"]
        self.channels = list(dic["channels"])
        self.stride_s = list(dic["stride_s"])
        self.stride_t = list(dic["stride_t"])
        assert len(self.channels) - 1 == len(self.stride_t) == len(self.stride_s) == 4
        self.conv1 = _StemConv(self.channels[0])
        self.norm1 = AffineParams(self.channels[0])
        inplanes = self.channels[0]
        layers = []
        for i, ch in enumerate(self.channels[1:]):
            down = None
            if self.stride_s[i] != 1 or inplanes != ch:                      # resnet3D.py:180
                down = nn.Sequential(ConvParams(inplanes, ch, 3, 3, bias=False), AffineParams(ch))
            layers.append(nn.Sequential(BasicBlock(inplanes, ch, self.stride_s[i], self.stride_t[i], down), BasicBlock(ch, ch)))
            inplanes = ch
        self.layer = nn.Sequential(*layers)
        self.conv_mu = ConvParams(self.channels[-1], self.z_dim, 4, 2, bias=True)
        self.conv_var = ConvParams(self.channels[-1], self.z_dim, 4, 2, bias=True)

    def _build_native(self):
        h = native.NativeEncoder3D(self.z_dim, self.channels, self.stride_s, self.stride_t, device=self.module_device())
        h.load(self.state_dict())
        return h

    def forward(self, x):
        """x [B,3,T,H,W] (or [B,T,3,H,W], transposed like the reference when dim 1 > dim 2) -> (sample, mu, logvar);
        sample = eps * exp(0.5 logvar) + mu with eps drawn on the CPU generator (resnet3D.py:199-203)."""
        if x.size(1) > x.size(2):
            x = x.transpose(1, 2)
        x = x.contiguous()
        eps = torch.randn(x.size(0), self.z_dim).to(x.device)
        return self.native().forward(x, eps)
