"""Mirror of the reference's ``stage2_cINN/modules/modules.py`` class surface (BasicFullyConnectedNet,
ActNorm) on top of libi2v_hip.so.  Same constructor signatures and state_dict keys; the arithmetic is
done by HIP kernels (see csrc/i2v_flow.hip), never by torch ops."""
import torch
import torch.nn as nn

import i2v_native as native
from i2v_params import LinearParams, NativeBacked, Slot


class BasicFullyConnectedNet(NativeBacked):
    """Linear(dim, hidden) -> LeakyReLU(0.01) -> [Linear(hidden, hidden) -> LeakyReLU(0.01)] x depth ->
    Linear(hidden, out)   (reference modules.py:9-30).  ``use_tanh`` / ``use_bn`` are never enabled on
    the sampling path (flow_blocks.py:68-70) and are rejected here."""

    def __init__(self, dim, depth, hidden_dim=256, use_tanh=False, use_bn=False, out_dim=None):
        super().__init__()
        if use_tanh or use_bn:
            raise NotImplementedError("BasicFullyConnectedNet: use_tanh/use_bn are not used by the cINN sampling path")
        self.dim, self.depth, self.hidden_dim = dim, depth, hidden_dim
        self.out_dim = dim if out_dim is None else out_dim
        layers = [LinearParams(dim, hidden_dim), Slot()]
        for _ in range(depth):
            layers += [LinearParams(hidden_dim, hidden_dim), Slot()]
        layers.append(LinearParams(hidden_dim, self.out_dim))
        self.main = nn.Sequential(*layers)

    def _build_native(self):
        h = native.NativeMLP(self.dim, self.hidden_dim, self.depth, self.out_dim, device=self.module_device())
        h.load({k: v for k, v in self.state_dict().items()})
        return h

    def forward(self, x):
        return self.native().forward(x.contiguous())


class ActNorm(NativeBacked):
    """h = scale * (x + loc), logdet = H*W*sum(log|scale|); reverse h = x/scale - loc (modules.py:33-104).
    Quirk Q1 is kept: with ``initialized == 0`` the first forward initialises loc/scale from the batch
    even in eval mode (modules.py:76-78)."""

    def __init__(self, num_features, logdet=False, affine=True):
        assert affine
        super().__init__()
        self.logdet = logdet
        self.loc = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.scale = nn.Parameter(torch.ones(1, num_features, 1, 1))
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))

    def initialize(self, input):
        # one-time data-dependent init (modules.py:43-63): per-channel mean and unbiased std over (B,H,W)
        with torch.no_grad():
            flat = input.permute(1, 0, 2, 3).contiguous().view(input.shape[1], -1)
            mean, std = native.channel_mean_std(flat)
            self.loc.data.copy_((-mean).view_as(self.loc))
            self.scale.data.copy_((1.0 / (std + 1e-6)).view_as(self.scale))
        self.refresh_native()

    def forward(self, input, reverse=False):
        if reverse:
            return self.reverse(input)
        squeeze = input.dim() == 2
        if squeeze:
            input = input[:, :, None, None]
        _, _, height, width = input.shape
        if self.initialized.item() == 0:
            self.initialize(input)
            self.initialized.fill_(1)
        h = native.actnorm(input.contiguous(), self.loc, self.scale, reverse=False)
        if squeeze:
            h = h.squeeze(-1).squeeze(-1)
        if self.logdet:
            logdet = native.actnorm_logdet(self.scale, height * width, input.shape[0])
            return h, logdet
        return h

    def reverse(self, output):
        squeeze = output.dim() == 2
        if squeeze:
            output = output[:, :, None, None]
        h = native.actnorm(output.contiguous(), self.loc, self.scale, reverse=True)
        if squeeze:
            h = h.squeeze(-1).squeeze(-1)
        return h
