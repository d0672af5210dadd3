"""Parameter holders shared by the module mirrors.

The reference keeps its weights in ``nn.Linear`` / ``nn.Conv*`` children; released checkpoints are
addressed by those children's ``state_dict`` keys.  Here the arithmetic happens in ``libi2v_hip.so``,
so the children are pure containers with the same parameter names, shapes and default initialisation
(kaiming-uniform(a=sqrt(5)) weights, uniform(+-1/sqrt(fan_in)) biases) and NO forward of their own.
"""
import math

import torch
import torch.nn as nn


def _init_weight(shape, fan_in):
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    return torch.empty(shape).uniform_(-bound, bound)


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError(f"{type(self).__name__} is a parameter container; the arithmetic runs in libi2v_hip.so "
                           "through the owning module's forward")


class LinearParams(_NoForward):
    """weight [out, in], bias [out] -- the state_dict footprint of nn.Linear."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(_init_weight((out_features, in_features), in_features))
        self.bias = nn.Parameter(_init_weight((out_features,), in_features))


class Slot(_NoForward):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the reference
    (activation layers sit at the odd indices of BasicFullyConnectedNet.main, modules.py:14-24)."""


class ConvParams(_NoForward):
    """State_dict footprint of nn.ConvNd, optionally wrapped by torch.nn.utils.spectral_norm
    (weight_orig Parameter + weight_u / weight_v buffers; decoder.py:20-25)."""

    def __init__(self, in_ch, out_ch, ksize, ndim, bias=True, spectral=False):
        super().__init__()
        shape = (out_ch, in_ch) + (ksize,) * ndim
        fan_in = in_ch * ksize ** ndim
        w = _init_weight(shape, fan_in)
        self.spectral = spectral
        if spectral:
            self.weight_orig = nn.Parameter(w)
            u = torch.randn(out_ch)
            v = torch.randn(fan_in)
            self.register_buffer("weight_u", u / (u.norm() + 1e-12))
            self.register_buffer("weight_v", v / (v.norm() + 1e-12))
        else:
            self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(_init_weight((out_ch,), fan_in))
        else:
            self.register_parameter("bias", None)


class AffineParams(_NoForward):
    """weight/bias [C] -- the footprint of nn.GroupNorm(affine=True)."""

    def __init__(self, num_features):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))


def _adopt(parent, value):
    """Record ``parent`` as the NativeBacked ancestor of the NativeBacked modules reachable from ``value`` (a module, or a
    container such as nn.ModuleList / nn.Sequential) that have no recorded ancestor yet (a nested owner -- a GeneratorBlock
    inside the Generator -- has already adopted its own children when it was constructed)."""
    if isinstance(value, nn.Module):
        for m in value.modules():
            if m is not parent and isinstance(m, NativeBacked) and m.__dict__.get("_native_parent") is None:
                object.__setattr__(m, "_native_parent", parent)


class NativeBacked(nn.Module):
    """Mixin: lazily builds the native handle from the module's own state_dict and drops it whenever the
    parameters may have changed (load_state_dict, .to()/.cuda(), explicit refresh_native())."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_native", None)
        object.__setattr__(self, "_native_parent", None)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh_native())

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        _adopt(self, value)

    def refresh_native(self):
        """Drop this module's handle and those of every NativeBacked ancestor that packed this module's parameters into its
        own handle (e.g. ``gen.g_0.load_state_dict(...)`` must invalidate ``gen``'s decoder handle too)."""
        m = self
        while m is not None:
            m._drop_native()
            m = m.__dict__.get("_native_parent")

    def _drop_native(self):
        """Forget this module's own handle (subclasses add what else depends on the parameters)."""
        object.__setattr__(self, "_native", None)

    def module_device(self):
        """Device the parameters live on (the device the native handle is created on)."""
        for t in self.parameters():
            return t.device
        for t in self.buffers():
            return t.device
        return None

    def _apply(self, fn, *a, **k):
        self.refresh_native()
        return super()._apply(fn, *a, **k)

    def _build_native(self):  # pragma: no cover - abstract
        raise NotImplementedError

    def native(self):
        if self._native is None:
            object.__setattr__(self, "_native", self._build_native())
        return self._native
