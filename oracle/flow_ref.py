"""CPU oracle for the cINN flow -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (torch CPU fp32, no nn.Module) of the reference's
``ConditionalFlow`` for parity checking of the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity pin: checked against golden vectors produced by the reference's own modules
(``tests/golden/make_golden.py`` -> ``tests/golden/flow_*.npz``), see
``tests/test_oracle_golden.py``.

Every function cites the reference lines (relative to /root/reference) it restates.
``sd`` is a ``ConditionalFlow.state_dict()``-shaped mapping {key: torch.Tensor}.
"""
import contextlib

import torch
import torch.nn.functional as F

# Emulation of the HIP path's optional fp16-operand mode (i2v_flow_cfg.linear_f16, BASELINE configs[4]): the operands of every
# Linear of the s- / t-nets are rounded to fp16 (round-to-nearest-even), products and sums stay fp32 -- NOT what the reference
# computes; used by the tests to pin that mode tightly, next to the looser comparison with the fp32 path.
_LINEAR_F16 = False


@contextlib.contextmanager
def linear_f16_emulation(on=True):
    global _LINEAR_F16
    old, _LINEAR_F16 = _LINEAR_F16, bool(on)
    try:
        yield
    finally:
        _LINEAR_F16 = old


def mlp(sd, prefix, x, depth=2):
    """BasicFullyConnectedNet.forward -- stage2_cINN/modules/modules.py:9-30.
    Linear -> LeakyReLU() [slope 0.01, modules.py:17,22] x (depth+1), final Linear."""
    h = x
    for li in range(depth + 2):
        w = sd[f"{prefix}main.{2 * li}.weight"]
        if _LINEAR_F16:
            h, w = h.half().float(), w.half().float()
        h = F.linear(h, w, sd[f"{prefix}main.{2 * li}.bias"])
        if li < depth + 1:
            h = F.leaky_relu(h, 0.01)
    return h


def actnorm_forward(sd, prefix, x):
    """ActNorm.forward -- modules.py:80-89 (H = W = 1): h = scale*(x+loc);
    logdet = sum(log|scale|) broadcast over the batch."""
    scale = sd[prefix + "scale"].reshape(1, -1)
    loc = sd[prefix + "loc"].reshape(1, -1)
    h = scale * (x + loc)
    logdet = torch.sum(torch.log(torch.abs(scale))) * torch.ones(x.shape[0], dtype=x.dtype)
    return h, logdet


def actnorm_reverse(sd, prefix, x):
    """ActNorm.reverse -- modules.py:93-104: h = x/scale - loc."""
    return x / sd[prefix + "scale"].reshape(1, -1) - sd[prefix + "loc"].reshape(1, -1)


def actnorm_data_init(x):
    """ActNorm.initialize -- modules.py:43-63 (quirk Q1: runs on the first forward whenever
    initialized == 0, also in eval mode).  Returns (loc, scale) as [C] tensors:
    loc = -mean_c, scale = 1/(std_c + 1e-6) with the unbiased std over the batch."""
    mean = x.mean(0)
    std = x.std(0)
    return -mean, 1.0 / (std + 1e-6)


def inv_lrelu_forward(x, alpha=0.9):
    """InvLeakyRelu.forward -- flow_blocks.py:176-182.  Quirk Q2: reported log-det is 0.0."""
    scaling = (x >= 0).to(x) + (x < 0).to(x) * alpha
    return x * scaling


def inv_lrelu_reverse(x, alpha=0.9):
    """InvLeakyRelu.reverse -- flow_blocks.py:184-187."""
    scaling = (x >= 0).to(x) + (x < 0).to(x) * alpha
    return x / scaling


def coupling_forward(sd, prefix, x, xc, mode="normal", depth=2):
    """ConditionalDoubleVectorCouplingBlock.forward (not reverse) -- flow_blocks.py:82-95.
    x [B,64], xc [B,E] -> (x' [B,64], logdet [B])."""
    logdet = torch.zeros(x.shape[0], dtype=x.dtype)
    for i in range(2):
        if i % 2 != 0:
            a, b = torch.chunk(x, 2, dim=1)
            x = torch.cat((b, a), dim=1)
        keep, apply = torch.chunk(x, 2, dim=1)  # x[idx_apply=0] feeds the nets, x[idx_keep=1] is transformed
        cin = torch.cat((keep, xc), dim=1) if mode == "normal" else xc
        s = mlp(sd, f"{prefix}s.{i}.", cin, depth)
        t = mlp(sd, f"{prefix}t.{i}.", cin, depth)
        x = torch.cat((keep, apply * s.exp() + t), dim=1)
        logdet = logdet + s.sum(dim=1)
    return x, logdet


def coupling_reverse(sd, prefix, x, xc, mode="normal", depth=2):
    """ConditionalDoubleVectorCouplingBlock.forward(reverse=True) -- flow_blocks.py:96-105."""
    for i in (1, 0):
        if i % 2 == 0:
            a, b = torch.chunk(x, 2, dim=1)
            x = torch.cat((b, a), dim=1)
        keep, apply = torch.chunk(x, 2, dim=1)
        cin = torch.cat((keep, xc), dim=1) if mode == "normal" else xc
        s = mlp(sd, f"{prefix}s.{i}.", cin, depth)
        t = mlp(sd, f"{prefix}t.{i}.", cin, depth)
        x = torch.cat((keep, (apply - t) * s.neg().exp()), dim=1)
    return x


def block_forward(sd, prefix, x, xc, mode="normal", depth=2, activation="lrelu"):
    """ConditionalFlatDoubleCouplingFlowBlock.forward -- flow_blocks.py:118-129:
    ActNorm -> activation -> coupling -> Shuffle, log-dets summed."""
    h, logdet = actnorm_forward(sd, prefix + "norm_layer.", x)
    if activation == "lrelu":
        h = inv_lrelu_forward(h)
    h, ld = coupling_forward(sd, prefix + "coupling.", h, xc, mode, depth)
    logdet = logdet + ld
    h = h[:, sd[prefix + "shuffle.forward_shuffle_idx"]]  # Shuffle.forward, flow_blocks.py:152
    return h, logdet


def block_reverse(sd, prefix, x, xc, mode="normal", depth=2, activation="lrelu"):
    """ConditionalFlatDoubleCouplingFlowBlock.forward(reverse=True) -- flow_blocks.py:130-136."""
    h = x[:, sd[prefix + "shuffle.backward_shuffle_idx"]]  # flow_blocks.py:154
    h = coupling_reverse(sd, prefix + "coupling.", h, xc, mode, depth)
    if activation == "lrelu":
        h = inv_lrelu_reverse(h)
    return actnorm_reverse(sd, prefix + "norm_layer.", h)


def _block_mode(fl, control):
    return "cond" if (fl % 4 != 0 and control) else "normal"  # flow_blocks.py:24


def flow_forward(sd, x, embedding, n_flows=20, depth=2, control=False):
    """ConditionalFlow.forward(reverse=False) -- flow_blocks.py:42-51 with
    conditioning_option "None" (get_model.py:40): the same embedding feeds every block.
    Returns (z~ [B,64,1,1], logdet [B]) -- the 4-D shape is what the reference returns."""
    x = x.reshape(x.shape[0], -1)
    logdet = torch.zeros(x.shape[0], dtype=x.dtype)
    for fl in range(n_flows):
        x, ld = block_forward(sd, f"sub_layers.{fl}.", x, embedding, _block_mode(fl, control), depth)
        logdet = logdet + ld
    return x[:, :, None, None], logdet


def flow_reverse(sd, x, embedding, n_flows=20, depth=2, control=False):
    """ConditionalFlow.forward(reverse=True) -- flow_blocks.py:53-57.  Returns [B,64,1,1]."""
    x = x.reshape(x.shape[0], -1)
    for fl in reversed(range(n_flows)):
        x = block_reverse(sd, f"sub_layers.{fl}.", x, embedding, _block_mode(fl, control), depth)
    return x[:, :, None, None]


def embed_pos(pos, cond_size=10):
    """SupervisedTransformer.embed_pos -- stage2_cINN/modules/INN.py:49-57: three one-hots of
    ``cond_size`` bins at index floor(pos*cond_size - 1e-4)."""
    p = pos * cond_size - 1e-4
    out = torch.zeros(pos.shape[0], 3 * cond_size)
    rows = torch.arange(pos.shape[0])
    for j in range(3):
        out[rows, j * cond_size + p[:, j].long()] = 1
    return out
