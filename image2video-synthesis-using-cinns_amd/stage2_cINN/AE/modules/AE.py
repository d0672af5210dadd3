"""Mirror of the conditioning encoder of the reference's ``stage2_cINN/AE/modules/AE.py`` (``ResnetEncoder``, lines 91-166;
row N1 of the coverage contract): torchvision-0.8.1 ResNet-50 with ``norm_layer`` InstanceNorm2d ("in") or BatchNorm2d
("bn", eval), ``fc`` replaced by ``DenseEncoderLayer`` = Conv2d(2048, 2*z_dim, 1).  The modules below only carry the
parameters under the reference's state_dict keys (``model.conv1.weight``, ``model.layer{k}.{i}.conv{j}.weight``,
``model.layer{k}.0.downsample.{0,1}.*``, ``model.bn1.*`` ..., ``model.fc.sub_layers.0.{weight,bias}``); the arithmetic runs
in libi2v_hip.so (csrc/i2v_embed.hip).  torchvision itself is not needed."""
import torch
import torch.nn as nn

import i2v_native as native
from i2v_params import ConvParams, NativeBacked, _NoForward
from stage2_cINN.AE.modules.distributions import PosteriorMean


class _BNParams(_NoForward):
    """State_dict footprint of nn.BatchNorm2d."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _INParams(_NoForward):
    """nn.InstanceNorm2d(planes): affine=False, track_running_stats=False -> no parameters, no buffers."""

    def __init__(self, c):
        super().__init__()


class _Bottleneck(_NoForward):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 1, 2, bias=False)
        self.bn1 = norm(planes)
        self.conv2 = ConvParams(planes, planes, 3, 2, bias=False)
        self.bn2 = norm(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1, 2, bias=False)
        self.bn3 = norm(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(ConvParams(inplanes, planes * 4, 1, 2, bias=False), norm(planes * 4))


class _ResNet50(_NoForward):
    def __init__(self, norm):
        super().__init__()
        self.conv1 = ConvParams(3, 64, 7, 2, bias=False)
        self.bn1 = norm(64)
        inplanes = 64
        for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
            layers = []
            for i in range(blocks):
                layers.append(_Bottleneck(inplanes, planes, 2 if (i == 0 and li > 1) else 1, i == 0, norm))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layers))


class DenseEncoderLayer(_NoForward):
    """Conv2d(in_channels, out_size, kernel=spatial_size) in a ModuleList (reference AE.py:54-81)."""

    def __init__(self, scale, spatial_size, out_size, in_channels=None, width_multiplier=1):
        super().__init__()
        self.in_channels = in_channels if in_channels is not None else int(width_multiplier * 64 * min(2 ** (scale - 1), 16))
        self.out_channels = out_size
        self.kernel_size = spatial_size
        self.sub_layers = nn.ModuleList([ConvParams(self.in_channels, out_size, spatial_size, 2, bias=True)])


_norm_options = {"in": _INParams, "bn": _BNParams}


class ResnetEncoder(NativeBacked):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.z_dim = config["z_dim"]
        self.be_deterministic = config["deterministic"]
        self.type = config["encoder_type"]
        if self.type != "resnet50":
            raise NotImplementedError("ResnetEncoder: every shipped config uses encoder_type 'resnet50'")
        if config["norm"] not in _norm_options:
            raise NotImplementedError(f"ResnetEncoder: norm '{config['norm']}' (shipped configs use 'in' or 'bn')")
        self.norm = config["norm"]
        self.model = _ResNet50(_norm_options[self.norm])
        # avgpool leaves a 1x1 map for every in_size, so the replaced fc is a 1x1 conv (AE.py:116-124)
        self.model.fc = DenseEncoderLayer(0, spatial_size=1, out_size=2 * self.z_dim, in_channels=2048)

    def _build_native(self):
        h = native.NativeEmbedder(self.z_dim, self.norm == "bn", device=self.module_device())
        h.load({k: v for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")})
        return h

    def forward(self, x):
        raise NotImplementedError("only encode(x).mode() -- the posterior mean -- is evaluated on the sampling path")

    def encode(self, input):
        mean = self.native().forward(input.contiguous())
        return PosteriorMean(mean[:, :, None, None])
