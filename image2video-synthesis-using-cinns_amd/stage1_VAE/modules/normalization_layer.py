"""Mirror of the reference's ``stage1_VAE/modules/normalization_layer.py`` class surface (Spade, ADAIN, Norm3D).

Inside ``Generator`` these modules are parameter containers: the whole decoder runs as one native call
(csrc/i2v_dec.hip).  Called on their own they run the same HIP kernels through ``i2v_norm_*``.
Inputs/outputs keep the reference layout ``[B, C, T, H, W]``."""
import torch.nn as nn

import i2v_native as native
from i2v_params import AffineParams, ConvParams, LinearParams, NativeBacked


class Spade(NativeBacked):
    """GroupNorm(G, C, affine=False)(x) * (1 + gamma(y)) + beta(y), y = start frame resized bilinearly
    (align_corners=True) -> Conv2d(3,128,3) -> LeakyReLU(0.2) -> Conv2d(128,C,3) x2  (reference :5-24)."""

    def __init__(self, num_features, num_groups=16):
        super().__init__()
        self.num_features = num_features
        while self.num_features % num_groups != 0:
            num_groups -= 1
        self.num_groups = num_groups
        self.conv = ConvParams(3, 128, 3, 2)
        self.conv_gamma = ConvParams(128, num_features, 3, 2)
        self.conv_beta = ConvParams(128, num_features, 3, 2)

    def _build_native(self):
        h = native.NativeNorm("spade", self.num_features, 0, device=self.module_device())
        h.load(self.state_dict())
        return h

    def forward(self, x, y):
        return self.native().forward(x.contiguous(), y.contiguous())


class Norm3D(NativeBacked):
    """GroupNorm(16, C, affine=True) on [B,C,T,H,W] (reference :27-35)."""

    def __init__(self, num_features, num_groups=16):
        super().__init__()
        if num_groups != 16:
            raise NotImplementedError("Norm3D: the decoder only uses 16 groups")
        self.num_features = num_features
        self.bn = AffineParams(num_features)

    def _build_native(self):
        h = native.NativeNorm("norm3d", self.num_features, 0, device=self.module_device())
        h.load(self.state_dict())
        return h

    def forward(self, x):
        return self.native().forward(x.contiguous(), None)


class ADAIN(NativeBacked):
    """gamma(z) * InstanceNorm3d(x) + beta(z), (gamma, beta) = Linear(z_dim, 2C)(z).chunk(2)  (reference :38-51)."""

    def __init__(self, num_features, z_dim):
        super().__init__()
        self.num_features = num_features
        self.z_dim = z_dim
        self.linear = LinearParams(z_dim, num_features * 2)

    def _build_native(self):
        h = native.NativeNorm("adain", self.num_features, self.z_dim, device=self.module_device())
        h.load(self.state_dict())
        return h

    def forward(self, x, y):
        return self.native().forward(x.contiguous(), y.contiguous())
