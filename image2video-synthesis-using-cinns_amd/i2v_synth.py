"""Deterministic synthetic weights for the cINN flow and the stage-1 decoder.

No released checkpoints are reachable from this build, so every
parity test, fixture and benchmark runs on weights produced here.  The routine is
numpy-only (``default_rng(seed)``), so the very same tensors can be loaded into

* the reference's own ``torch.nn`` modules (``tests/golden/make_golden.py``, run
  once in the build container), and
* this package's modules / the CPU oracle (everywhere else).

Keys and shapes follow the reference ``state_dict`` layout:
``ConditionalFlow`` (stage2_cINN/modules/flow_blocks.py:8-57) and ``Generator``
(stage1_VAE/modules/decoder.py:55-83).

Requirements the synthesiser meets (found by running the reference):
  * spectral-norm ``weight_u/weight_v`` are power-iterated so that
    sigma = u^T W v ~ sigma_max > 0, as in a trained checkpoint
    (``negative_sigma`` flips the sign of ``u`` for selected convs to pin the
    signed-sigma quirk D6);
  * ActNorm ``scale`` has negative entries and ``loc != 0``, ``initialized = 1``;
  * the last Linear of every s-net is scaled so |s| stays O(0.1..1): the
    reference has no clamp in front of ``exp`` (flow_blocks.py:91);
  * ``conv_img`` is scaled so the pre-tanh values are O(0.5).
"""
import numpy as np

__all__ = ["flow_state_dict", "decoder_state_dict", "embedder_state_dict", "encoder3d_state_dict", "bench_inputs"]


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def flow_state_dict(seed=7, n_flows=20, in_channels=64, embedding_dim=64, hidden_dim=512,
                    hidden_depth=2, control=False, s_last_gain=0.1, hidden_gain=1.7):
    """state_dict of ``ConditionalFlow`` as {key: np.ndarray}.

    ``control=True`` reproduces flow_blocks.py:24: blocks with fl % 4 != 0 use
    mode 'cond' (first Linear sees only the embedding).
    """
    rng = np.random.default_rng(seed)
    half = in_channels // 2
    sd = {}
    for fl in range(n_flows):
        p = f"sub_layers.{fl}."
        mode_cond = control and (fl % 4 != 0)
        dim = embedding_dim if mode_cond else half + embedding_dim
        sd[p + "norm_layer.loc"] = (0.1 * rng.standard_normal((1, in_channels, 1, 1))).astype(np.float32)
        scale = np.exp(0.2 * rng.standard_normal((1, in_channels, 1, 1)))
        sign = np.where(rng.uniform(size=scale.shape) < 0.25, -1.0, 1.0)
        sd[p + "norm_layer.scale"] = (scale * sign).astype(np.float32)
        sd[p + "norm_layer.initialized"] = np.array(1, dtype=np.uint8)
        for net in ("s", "t"):
            for i in range(2):
                q = f"{p}coupling.{net}.{i}.main."
                dims = [dim] + [hidden_dim] * (hidden_depth + 1) + [half]
                for li in range(hidden_depth + 2):
                    fan_in, fan_out = dims[li], dims[li + 1]
                    last = li == hidden_depth + 1
                    gain = hidden_gain
                    if last:
                        gain = s_last_gain if net == "s" else 1.0
                    bound = gain / np.sqrt(fan_in)
                    sd[f"{q}{2 * li}.weight"] = _uniform(rng, (fan_out, fan_in), bound)
                    sd[f"{q}{2 * li}.bias"] = _uniform(rng, (fan_out,), 0.1 if not last else 0.05)
        idx = rng.permutation(in_channels).astype(np.int64)
        sd[p + "shuffle.forward_shuffle_idx"] = idx
        sd[p + "shuffle.backward_shuffle_idx"] = np.argsort(idx).astype(np.int64)
    return sd


def _power_iterate(w_mat, rng, iters=20):
    """Converged spectral-norm vectors, as torch's hook leaves them after training
    (torch.nn.utils.spectral_norm: u = normalize(W v), v = normalize(W^T u))."""
    w = w_mat.astype(np.float64)
    u = rng.standard_normal(w.shape[0])
    u /= np.linalg.norm(u) + 1e-12
    v = rng.standard_normal(w.shape[1])
    v /= np.linalg.norm(v) + 1e-12
    for _ in range(iters):
        v = w.T @ u
        v /= np.linalg.norm(v) + 1e-12
        u = w @ v
        u /= np.linalg.norm(u) + 1e-12
    return u.astype(np.float32), v.astype(np.float32)


def _sn_conv(sd, rng, name, cout, cin, k, bias=True, negative=False):
    fan_in = cin * k ** 3
    w = _uniform(rng, (cout, cin, k, k, k), 1.0 / np.sqrt(fan_in))
    u, v = _power_iterate(w.reshape(cout, -1), rng)
    if negative:
        u = -u
    sd[name + ".weight_orig"] = w
    sd[name + ".weight_u"] = u
    sd[name + ".weight_v"] = v
    if bias:
        sd[name + ".bias"] = _uniform(rng, (cout,), 0.1)


def _plain_conv3d(sd, rng, name, cout, cin, k, bias=True):
    fan_in = cin * k ** 3
    sd[name + ".weight"] = _uniform(rng, (cout, cin, k, k, k), 1.0 / np.sqrt(fan_in))
    if bias:
        sd[name + ".bias"] = _uniform(rng, (cout,), 0.1)


def _block(sd, rng, name, n_in, n_out, z_dim, spectral, negative_sigma):
    n_mid = min(n_in, n_out)
    conv = _sn_conv if spectral else None
    for cname, co, ci, k, b in (("conv_0", n_mid, n_in, 3, True), ("conv_1", n_out, n_mid, 3, True)):
        full = f"{name}.{cname}"
        if spectral:
            conv(sd, rng, full, co, ci, k, bias=b, negative=full in negative_sigma)
        else:
            _plain_conv3d(sd, rng, full, co, ci, k, bias=b)
    if n_in != n_out:
        full = f"{name}.conv_s"
        if spectral:
            _sn_conv(sd, rng, full, n_out, n_in, 1, bias=False, negative=full in negative_sigma)
        else:
            _plain_conv3d(sd, rng, full, n_out, n_in, 1, bias=False)
    # Spade (normalization_layer.py:13-15): Conv2d(3,128), Conv2d(128,C) x2
    sd[f"{name}.norm_0.conv.weight"] = _uniform(rng, (128, 3, 3, 3), 1.5 / np.sqrt(27))
    sd[f"{name}.norm_0.conv.bias"] = _uniform(rng, (128,), 0.2)
    for gb in ("conv_gamma", "conv_beta"):
        sd[f"{name}.norm_0.{gb}.weight"] = _uniform(rng, (n_in, 128, 3, 3), 1.0 / np.sqrt(128 * 9))
        sd[f"{name}.norm_0.{gb}.bias"] = _uniform(rng, (n_in,), 0.1)
    # ADAIN (normalization_layer.py:44): Linear(z_dim, 2C); gamma multiplies directly (no 1+)
    w = _uniform(rng, (2 * n_mid, z_dim), 0.5 / np.sqrt(z_dim))
    b = _uniform(rng, (2 * n_mid,), 0.2)
    b[:n_mid] += 1.0
    sd[f"{name}.norm_1.linear.weight"] = w
    sd[f"{name}.norm_1.linear.bias"] = b
    if n_in != n_out:  # Norm3D (normalization_layer.py:31): GroupNorm affine
        sd[f"{name}.norm_s.bn.weight"] = (1.0 + _uniform(rng, (n_in,), 0.2)).astype(np.float32)
        sd[f"{name}.norm_s.bn.bias"] = _uniform(rng, (n_in,), 0.2)


def decoder_state_dict(seed=7, channel_factor=64, z_dim=64, spectral_norm=True, negative_sigma=(), conv_img_gain=1.0):
    """state_dict of ``Generator`` (decoder.py:55-83) as {key: np.ndarray}.

    ``conv_img_gain``: factor on ``conv_img`` (weight and bias) applied AFTER the draw, so the random stream -- and every
    other tensor -- is the one of gain 1; fixtures whose latents come out of the cINN (|z| larger than a unit normal) use it
    to keep the frames out of tanh saturation.

    ``negative_sigma``: iterable of conv names (e.g. "g_1.conv_0") whose ``weight_u`` is
    negated so that sigma < 0 (quirk D6: the reference divides by the signed value).
    """
    rng = np.random.default_rng(seed)
    nf = channel_factor
    negative_sigma = set(negative_sigma)
    sd = {}
    sd["fc.weight"] = _uniform(rng, (16 * 16 * nf, z_dim), 1.0 / np.sqrt(z_dim))
    sd["fc.bias"] = _uniform(rng, (16 * 16 * nf,), 0.1)
    plan = (("head_0", 16, 16), ("g_0", 16, 16), ("g_1", 16, 8), ("g_2", 8, 4), ("g_3", 4, 2), ("g_4", 2, 1))
    for name, a, b in plan:
        _block(sd, rng, name, a * nf, b * nf, z_dim, spectral_norm, negative_sigma)
    fan_in = nf * 27
    sd["conv_img.weight"] = _uniform(rng, (3, nf, 3, 3, 3), 1.2 / np.sqrt(fan_in))
    sd["conv_img.bias"] = _uniform(rng, (3,), 0.1)
    if conv_img_gain != 1.0:
        sd["conv_img.weight"] = (sd["conv_img.weight"] * np.float32(conv_img_gain)).astype(np.float32)
        sd["conv_img.bias"] = (sd["conv_img.bias"] * np.float32(conv_img_gain)).astype(np.float32)
    return sd


def embedder_state_dict(seed=7, z_dim=64, norm="in"):
    """state_dict of ``ResnetEncoder`` (stage2_cINN/AE/modules/AE.py:91-124: torchvision-0.8.1 resnet50 with the given
    norm layer, fc = Conv2d(2048, 2*z_dim, 1)) as {key: np.ndarray}; He-style weights so activations stay O(1)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = _uniform(rng, (cout, cin, k, k), np.sqrt(3.0 / (cin * k * k)))

    def bn(name, c):
        if norm != "bn":
            return
        sd[name + ".weight"] = (1.0 + _uniform(rng, (c,), 0.2)).astype(np.float32)
        sd[name + ".bias"] = _uniform(rng, (c,), 0.2)
        sd[name + ".running_mean"] = _uniform(rng, (c,), 0.3)
        sd[name + ".running_var"] = (1.0 + _uniform(rng, (c,), 0.5)).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(100, dtype=np.int64)

    conv("model.conv1", 64, 3, 7)
    bn("model.bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for i in range(blocks):
            p = f"model.layer{li}.{i}."
            conv(p + "conv1", planes, inplanes, 1)
            bn(p + "bn1", planes)
            conv(p + "conv2", planes, planes, 3)
            bn(p + "bn2", planes)
            conv(p + "conv3", planes * 4, planes, 1)
            bn(p + "bn3", planes * 4)
            if i == 0:
                conv(p + "downsample.0", planes * 4, inplanes, 1)
                bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    sd["model.fc.sub_layers.0.weight"] = _uniform(rng, (2 * z_dim, 2048, 1, 1), 1.0 / np.sqrt(2048))
    sd["model.fc.sub_layers.0.bias"] = _uniform(rng, (2 * z_dim,), 0.1)
    return sd


def encoder3d_state_dict(seed=7, z_dim=64, channels=(64, 128, 256, 512, 512), stride_s=(1, 2, 2, 2)):
    """state_dict of the motion ``Encoder`` (stage1_VAE/modules/resnet3D.py:138-219, resnet18 BasicBlocks) as
    {key: np.ndarray}."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, cout, cin, *k):
        fan_in = cin * int(np.prod(k))
        sd[name + ".weight"] = _uniform(rng, (cout, cin) + tuple(k), np.sqrt(3.0 / fan_in))

    def gn(name, c):
        sd[name + ".weight"] = (1.0 + _uniform(rng, (c,), 0.2)).astype(np.float32)
        sd[name + ".bias"] = _uniform(rng, (c,), 0.2)

    conv("conv1", channels[0], 3, 3, 7, 7)
    gn("norm1", channels[0])
    inplanes = channels[0]
    for L, ch in enumerate(channels[1:]):
        for i in range(2):
            p = f"layer.{L}.{i}."
            conv(p + "conv1", ch, inplanes if i == 0 else ch, 3, 3, 3)
            gn(p + "bn1", ch)
            conv(p + "conv2", ch, ch, 3, 3, 3)
            gn(p + "bn2", ch)
            if i == 0 and (stride_s[L] != 1 or inplanes != ch):
                conv(p + "downsample.0", ch, inplanes, 3, 3, 3)
                gn(p + "downsample.1", ch)
        inplanes = ch
    for head in ("conv_mu", "conv_var"):
        sd[head + ".weight"] = _uniform(rng, (z_dim, channels[-1], 4, 4), 1.0 / np.sqrt(channels[-1] * 16))
        sd[head + ".bias"] = _uniform(rng, (z_dim,), 0.1)
    return sd


def bench_inputs(batch, img_size, emb_dim, z_dim=64):
    """Synthetic inputs of SURVEY §8d, bit-stable across hosts: CPU torch generators with
    fixed seeds (x_0 1234, residual 4321, embed 2468).  Drawn for the GLOBAL batch; the
    multi-GPU harness slices them per rank."""
    import torch
    g = torch.Generator().manual_seed(1234)
    x0 = 2.0 * torch.rand(batch, 3, img_size, img_size, generator=g) - 1.0
    g = torch.Generator().manual_seed(4321)
    residual = torch.randn(batch, z_dim, generator=g)
    g = torch.Generator().manual_seed(2468)
    embed = torch.randn(batch, emb_dim, generator=g)
    return x0, residual, embed
