"""CPU oracle for the conditioning embedder (row N1) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

**Parity unpinned.**  The reference builds this network from ``torchvision.models.resnet50`` (pinned 0.8.1,
environment.yaml:11), which is NOT vendored under /root/reference and not installed in the build container, and the
reference has no tests or fixtures for it.  This file restates the published torchvision-0.8.1 ResNet-50 forward
(torchvision/models/resnet.py: Bottleneck with the stride on conv2, expansion 4, layers [3,4,6,3], conv1 7x7/2 + maxpool
3x3/2, AdaptiveAvgPool2d(1)) as the reference calls it: ``ResnetEncoder.features`` / ``forward`` / ``encode(...).mode()``
(stage2_cINN/AE/modules/AE.py:126-166, distributions.py:9,41-42), with norm_layer InstanceNorm2d(planes) [affine=False,
no running stats] or eval-mode BatchNorm2d, and fc = Conv2d(2048, 2E, 1).  It anchors the HIP embedder's parity test; it
has not been checked against reference outputs."""
import torch
import torch.nn.functional as F


def _norm(sd, name, x, kind):
    if kind == "in":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=False, eps=1e-5)


def _bottleneck(sd, p, x, stride, has_down, kind):
    out = F.relu(_norm(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"]), kind))
    out = F.relu(_norm(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), kind))
    out = _norm(sd, p + "bn3", F.conv2d(out, sd[p + "conv3.weight"]), kind)
    identity = x
    if has_down:
        identity = _norm(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), kind)
    return F.relu(out + identity)


def encode_mode(sd, x, norm="in"):
    """ResnetEncoder.encode(x).mode() -> [B, E, 1, 1] (AE.py:126-141,163-166; the [-1,1] image is used as is)."""
    h = F.conv2d(x, sd["model.conv1.weight"], stride=2, padding=3)
    h = F.relu(_norm(sd, "model.bn1", h, norm))
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for li, blocks in enumerate((3, 4, 6, 3), start=1):
        for i in range(blocks):
            h = _bottleneck(sd, f"model.layer{li}.{i}.", h, 2 if (i == 0 and li > 1) else 1, i == 0, norm)
    h = F.adaptive_avg_pool2d(h, (1, 1))
    h = F.conv2d(h, sd["model.fc.sub_layers.0.weight"], sd["model.fc.sub_layers.0.bias"])
    mean, _logvar = torch.chunk(h, 2, dim=1)
    return mean
