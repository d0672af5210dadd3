"""Mirror of the reference's ``stage2_cINN/modules/flow_blocks.py`` class surface on top of libi2v_hip.so.

Constructor signatures, forward/reverse signatures, the 2-D vs 4-D ``[B,C,1,1]`` shape conventions and
the state_dict keys are those of the reference; every forward runs HIP kernels (csrc/i2v_flow.hip,
csrc/i2v_ops.hip).  A module on the CPU raises -- there is no eager fallback.
"""
import torch
import torch.nn as nn

import i2v_native as native
from i2v_params import NativeBacked
from stage2_cINN.modules.modules import ActNorm, BasicFullyConnectedNet


def _prefixed(module, prefix):
    return {prefix + k: v for k, v in module.state_dict().items()}


class ConditionalFlow(NativeBacked):
    """Flat conditional flow: n_flows x [ActNorm -> InvLeakyRelu -> double affine coupling -> Shuffle]
    (reference flow_blocks.py:8-60).  forward returns (z~ [B,C,1,1], logdet [B]); reverse returns z [B,C,1,1]."""

    def __init__(self, in_channels, embedding_dim, hidden_dim, hidden_depth, n_flows, conditioning_option="none",
                 activation="lrelu", control=False):
        super().__init__()
        self.in_channels = in_channels
        self.cond_channels = embedding_dim
        self.mid_channels = hidden_dim
        self.num_blocks = hidden_depth
        self.n_flows = n_flows
        self.conditioning_option = conditioning_option
        self.activation = activation
        self.control = bool(control)
        if conditioning_option.lower() != "none":
            # the sampling path always passes "None" (get_model.py:40); the per-block 1x1 conditioning convs of
            # the "parallel"/"sequential" options (flow_blocks.py:28-40) are outside the hot path
            raise NotImplementedError("ConditionalFlow: only conditioning_option='none' is supported")
        layers = []
        for fl in range(n_flows):
            mode = "cond" if (fl % 4 != 0 and control) else "normal"
            layers.append(ConditionalFlatDoubleCouplingFlowBlock(
                in_channels, embedding_dim, hidden_dim, hidden_depth, activation=activation, mode=mode))
        self.sub_layers = nn.ModuleList(layers)
        # the reference records every block's output / log-det of the last forward in these two lists
        # (flow_blocks.py:34-35,49-50); here that costs 20 separate block launches instead of one fused pass, so it is
        # opt-in: set ``record_intermediates = True`` (the lists stay empty otherwise)
        self.record_intermediates = False
        self._init_checked = False
        # Not a reference argument: operand precision of the s- / t-net Linear layers.  None = env I2V_FLOW_F16 (default 0: exact
        # fp32 matrix cores); 1 = fp16 operands, fp32 accumulation (BASELINE configs[4]'s "fp16 MFMA conditioning GEMM"; outside
        # the 1e-4 fp32 gate).  Set it before the first call (or call refresh_native()).
        self.linear_f16 = None

    def _build_native(self):
        h = native.NativeFlow(self.in_channels, self.cond_channels, self.mid_channels, self.num_blocks, self.n_flows,
                              control=1 if self.control else 0, activation=self.activation, device=self.module_device(),
                              linear_f16=self.linear_f16)
        h.load(self.state_dict())
        return h

    def _data_dependent_init(self, x, embedding):
        """Quirk Q1 (modules.py:76-78): ActNorms with initialized == 0 initialise themselves from their own
        input on the first forward, block after block, also in eval mode."""
        h = x
        for blk in self.sub_layers:
            h, _ = blk(h[:, :, None, None], embedding[:, :, None, None])
        object.__setattr__(self, "_native", None)   # loc / scale of the blocks changed: rebuild the handle

    def _drop_native(self):
        super()._drop_native()
        object.__setattr__(self, "_init_checked", False)  # parameters may have changed: look at `initialized` again

    def _forward_recorded(self, x2, e2):
        """Block-by-block forward that fills last_outs / last_logdets like flow_blocks.py:42-51."""
        h, logdet = x2[:, :, None, None], 0.0
        e4 = e2[:, :, None, None]
        for blk in self.sub_layers:
            h, ld = blk(h.reshape(h.shape[0], -1, 1, 1), e4)
            logdet = logdet + ld
            self.last_outs.append(h)              # [B, C] like the reference (the block's output is 2-D, flow_blocks.py:47-49)
            self.last_logdets.append(logdet)      # the RUNNING total, as flow_blocks.py:48,50 appends it
        return h[:, :, None, None], logdet

    def forward(self, x, embedding, reverse=False):
        self.last_outs, self.last_logdets = [], []
        x2 = x.reshape(x.shape[0], -1).contiguous()
        e2 = embedding.reshape(embedding.shape[0], -1).contiguous()
        if not reverse:
            # Q1: the `initialized` buffers are looked at on the first FORWARD after every (re)load / move, whether or not a
            # reverse pass has already built the native handle (no per-call device sync afterwards)
            if not self._init_checked:
                if any(int(b.norm_layer.initialized.item()) == 0 for b in self.sub_layers):
                    self._data_dependent_init(x2, e2)
                object.__setattr__(self, "_init_checked", True)
            if self.record_intermediates:
                return self._forward_recorded(x2, e2)
            out, logdet = self.native().forward(x2, e2)
            return out[:, :, None, None], logdet
        return self.native().inverse(x2, e2)[:, :, None, None]

    def reverse(self, out, xcond):
        return self(out, xcond, reverse=True)


class ConditionalDoubleVectorCouplingBlock(NativeBacked):
    """Two affine couplings with half swap (reference flow_blocks.py:63-105)."""

    def __init__(self, in_channels, cond_channels, hidden_dim, depth=2, mode="normal"):
        super().__init__()
        dim = in_channels // 2 + cond_channels if mode == "normal" else cond_channels
        self.s = nn.ModuleList([BasicFullyConnectedNet(dim=dim, depth=depth, hidden_dim=hidden_dim,
                                                       out_dim=in_channels // 2) for _ in range(2)])
        self.t = nn.ModuleList([BasicFullyConnectedNet(dim=dim, depth=depth, hidden_dim=hidden_dim,
                                                       out_dim=in_channels // 2) for _ in range(2)])
        self.mode = mode
        self._geom = (in_channels, cond_channels, hidden_dim, depth)

    def _build_native(self):
        c, e, hdim, depth = self._geom
        h = native.NativeFlow(c, e, hdim, depth, 1, control=2 if self.mode != "normal" else 0, activation="none",
                              skip_actnorm=True, skip_shuffle=True, device=self.module_device())
        h.load(_prefixed(self, "sub_layers.0.coupling."))
        return h

    def forward(self, x, xc, reverse=False):
        assert len(x.shape) == 4
        assert len(xc.shape) == 4
        x = x.squeeze(-1).squeeze(-1).contiguous()
        xc = xc.squeeze(-1).squeeze(-1).contiguous()
        if not reverse:
            return self.native().forward(x, xc)
        return self.native().inverse(x, xc)[:, :, None, None]


class ConditionalFlatDoubleCouplingFlowBlock(NativeBacked):
    """ActNorm -> activation -> coupling -> Shuffle, log-dets summed (reference flow_blocks.py:108-139)."""

    def __init__(self, in_channels, cond_channels, hidden_dim, hidden_depth, activation="lrelu", mode="normal"):
        super().__init__()
        possible = {"lrelu": InvLeakyRelu, "none": IgnoreLeakyRelu}
        self.norm_layer = ActNorm(in_channels, logdet=True)
        self.coupling = ConditionalDoubleVectorCouplingBlock(in_channels, cond_channels, hidden_dim, hidden_depth, mode)
        self.activation = possible[activation]()
        self.shuffle = Shuffle(in_channels)
        self._geom = (in_channels, cond_channels, hidden_dim, hidden_depth, activation, mode)

    def _build_native(self):
        c, e, hdim, depth, act, mode = self._geom
        h = native.NativeFlow(c, e, hdim, depth, 1, control=2 if mode != "normal" else 0, activation=act,
                              device=self.module_device())
        h.load(_prefixed(self, "sub_layers.0."))
        return h

    def forward(self, x, xcond, reverse=False):
        x2 = x.reshape(x.shape[0], -1).contiguous()
        e2 = xcond.reshape(xcond.shape[0], -1).contiguous()
        if not reverse:
            if int(self.norm_layer.initialized.item()) == 0:
                self.norm_layer(x2)  # Q1: initialises loc/scale from this batch
                self.refresh_native()
            return self.native().forward(x2, e2)
        return self.native().inverse(x2, e2)[:, :, None, None]

    def reverse(self, out, xcond):
        return self.forward(out, xcond, reverse=True)


class Shuffle(nn.Module):
    """Fixed random channel permutation, log-det 0 (reference flow_blocks.py:142-154) -- a 1x1 convolution with
    a permutation matrix, executed as a gather."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        idx = torch.randperm(in_channels)
        self.register_buffer("forward_shuffle_idx", nn.Parameter(idx, requires_grad=False))
        self.register_buffer("backward_shuffle_idx", nn.Parameter(torch.argsort(idx), requires_grad=False))

    def forward(self, x, reverse=False, conditioning=None):
        if not reverse:
            return native.gather_channels(x.contiguous(), self.forward_shuffle_idx), 0
        return native.gather_channels(x.contiguous(), self.backward_shuffle_idx)


class IgnoreLeakyRelu(nn.Module):
    """performs identity op. (reference flow_blocks.py:156-169)"""

    def forward(self, input, reverse=False):
        if reverse:
            return self.reverse(input)
        return input, 0.0

    def reverse(self, input):
        return input


class InvLeakyRelu(nn.Module):
    """x * (x >= 0 ? 1 : alpha); the forward log-det is reported as 0.0 (quirk Q2, reference flow_blocks.py:172-187)."""

    def __init__(self, alpha=0.9):
        super().__init__()
        self.alpha = alpha

    def forward(self, input, reverse=False):
        if reverse:
            return self.reverse(input)
        return native.inv_lrelu(input.contiguous(), self.alpha, reverse=False), 0.0

    def reverse(self, input):
        return native.inv_lrelu(input.contiguous(), self.alpha, reverse=True)
