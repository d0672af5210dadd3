"""ctypes binding of ``libi2v_hip.so`` (C ABI declared in ``include/i2v_hip.h``).

PyTorch-ROCm tensors are used only as containers: every call passes raw device pointers
(``tensor.data_ptr()``) and the current HIP stream.  There is NO CPU fallback: if the shared
library is missing, or a tensor does not live on a GPU, the call raises.
"""
import ctypes
import weakref
import functools
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

import numpy as np
import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libi2v_hip.so")
if os.environ.get("I2V_LIB_PATH"):   # measurement: another build of the same library (tools/build_measurement_libs.sh), relative to the repo
    LIB_PATH = os.path.join(os.path.dirname(_PKG), os.environ["I2V_LIB_PATH"])
CSRC = os.path.join(_PKG, "csrc")

I2V_F32, I2V_I64, I2V_U8 = 0, 1, 2


class I2VError(RuntimeError):
    pass


class _Tensor(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("numel", c_int64), ("dtype", c_int32)]


class FlowCfg(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("in_channels", "embedding_dim", "hidden_dim", "hidden_depth", "n_flows",
                                      "control", "activation", "skip_actnorm", "skip_shuffle", "use_graph", "linear_f16")]


class DecCfg(ctypes.Structure):
    _fields_ = [("channel_factor", c_int32), ("z_dim", c_int32), ("upsample_s", c_int32 * 2),
                ("upsample_t", c_int32 * 2), ("spectral_norm", c_int32), ("mma", c_int32)]


class Enc3dCfg(ctypes.Structure):
    _fields_ = [("z_dim", c_int32), ("channels", c_int32 * 5), ("stride_s", c_int32 * 4), ("stride_t", c_int32 * 4),
                ("use_max_pool", c_int32)]


_lib = None

# symbol -> (restype, argtypes); also the list the CPU test-suite checks the .so exports
SYMBOLS = {
    "i2v_last_error": (c_char_p, []),
    "i2v_version": (c_int32, []),
    "i2v_device_count": (c_int32, []),
    "i2v_flow_create": (c_int32, [POINTER(FlowCfg), POINTER(c_void_p)]),
    "i2v_flow_destroy": (None, [c_void_p]),
    "i2v_flow_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_flow_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "i2v_flow_param_bytes": (c_size_t, [c_void_p]),
    "i2v_flow_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_flow_inverse": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_mlp_create": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "i2v_mlp_destroy": (None, [c_void_p]),
    "i2v_mlp_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_mlp_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "i2v_mlp_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_channel_op": (c_int32, [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                 c_float, c_void_p]),
    "i2v_row_mean_std": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "i2v_actnorm_logdet": (c_int32, [c_void_p, c_int32, c_float, c_void_p, c_int32, c_void_p]),
    "i2v_probe_mfma_f16": (c_int32, [c_int32, c_int32, c_void_p, POINTER(c_double), c_void_p]),
    "i2v_gblock_create": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "i2v_gblock_destroy": (None, [c_void_p]),
    "i2v_gblock_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_gblock_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32, c_int32, c_int32]),
    "i2v_gblock_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                     c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "i2v_gblock_status": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_void_p]),
    "i2v_gblock_norm": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_int32,
                                  c_int32, c_int32, c_int32, c_void_p]),
    "i2v_embedder_create": (c_int32, [c_int32, c_int32, POINTER(c_void_p)]),
    "i2v_embedder_destroy": (None, [c_void_p]),
    "i2v_embedder_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_embedder_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32, c_int32]),
    "i2v_embedder_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_encoder3d_create": (c_int32, [POINTER(Enc3dCfg), POINTER(c_void_p)]),
    "i2v_encoder3d_destroy": (None, [c_void_p]),
    "i2v_encoder3d_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_encoder3d_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32, c_int32, c_int32]),
    "i2v_encoder3d_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_dec_create": (c_int32, [POINTER(DecCfg), POINTER(c_void_p)]),
    "i2v_dec_destroy": (None, [c_void_p]),
    "i2v_dec_load": (c_int32, [c_void_p, POINTER(_Tensor), c_int32]),
    "i2v_dec_out_shape": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "i2v_dec_workspace_bytes": (c_size_t, [c_void_p, c_int32, c_int32, c_int32]),
    "i2v_dec_flops_per_sample": (c_double, [c_void_p, c_int32, c_int32]),
    "i2v_dec_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_dec_forward_strided": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                          c_int32, c_void_p]),
    "i2v_dec_prepare": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_int32, c_void_p]),
    "i2v_dec_prepare_cancel": (c_int32, [c_void_p]),
    "i2v_dec_join": (c_int32, [c_void_p, c_void_p]),
    "i2v_dec_set_side_stream": (c_int32, [c_void_p, c_void_p]),
    "i2v_dec_fallback_layers": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "i2v_dec_set_profile": (c_int32, [c_void_p, c_int32]),
    "i2v_dec_debug_tap": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_size_t]),
    "i2v_dec_get_profile": (c_int32, [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "i2v_dec_status": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_void_p]),
    "i2v_dec_get_layer_profile": (c_int32, [c_void_p, c_int32, ctypes.c_char_p, c_int32, POINTER(c_double), POINTER(c_double),
                                            POINTER(c_double), POINTER(c_int64), POINTER(c_int32)]),
}


def build(force=False):
    """Compile ``libi2v_hip.so`` for gfx950 in-tree (``make`` in csrc/; hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j4"], check=True)
    if not os.path.exists(LIB_PATH):
        raise I2VError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


MEASURE_LIB_PATH = os.path.join(_PKG, "lib", "libi2v_hip_measure.so")


def build_measure():
    """Compile the MEASUREMENT build of the library (``make measure``: -DI2V_MEASURE, the F(4,3) kernel's structure switches and
    persistent instantiations).  Only tests and A/B runs load it, through ``I2V_LIB_PATH`` in a process of their own."""
    subprocess.run(["make", "-C", CSRC, "-j4", "measure"], check=True)
    if not os.path.exists(MEASURE_LIB_PATH):
        raise I2VError(f"build did not produce {MEASURE_LIB_PATH}")
    return MEASURE_LIB_PATH


def lib():
    """The loaded shared library.  Fails loudly when it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise I2VError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C {CSRC}`; this package has no CPU/eager fallback")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise I2VError(f"{what} failed ({rc}): {lib().i2v_last_error().decode(errors='replace')}")


def _require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise I2VError("libi2v_hip kernels need tensors on a HIP device (got a CPU tensor); "
                           "this package has no CPU fallback -- move the module and its inputs to 'cuda'")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise I2VError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _device_of(device):
    """torch.device of a handle: an explicit cuda device, else the current one.  A handle owns packed weights on ONE GPU."""
    if device is None:
        if not torch.cuda.is_available():
            raise I2VError("libi2v_hip needs a HIP device (torch.cuda.is_available() is False); this package has no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise I2VError(f"libi2v_hip kernels run on HIP devices only (got '{device}'); this package has no CPU fallback -- "
                       "move the module and its inputs to 'cuda'")
    return torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())


class _Handle:
    """Common part of the native handles: the device binding.  Creation, load and every call run with the handle's device
    current (``torch.cuda.device``), so the weights, the launches and the stream all belong to the GPU the tensors live on;
    tensors on another device are rejected (the C side checks the same thing and returns I2V_E_INVALID)."""

    def _bind(self, device):
        self.device = _device_of(device)
        return torch.cuda.device(self.device)

    def _on(self, *tensors):
        for t in tensors:
            if t is not None and t.is_cuda and t.device != self.device:
                raise I2VError(f"tensor on {t.device} passed to a handle that lives on {self.device}: a native handle serves one GPU "
                               "(move the module with .to(device) -- that rebuilds the handle -- or the input)")
        return torch.cuda.device(self.device)


def _pack_state_dict(sd):
    """{key: tensor/ndarray} -> (ctypes array of i2v_tensor, keep-alive list)."""
    keep, items = [], []
    for k, v in sd.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().contiguous().numpy()
        a = np.ascontiguousarray(v)
        if a.dtype == np.float32:
            dt = I2V_F32
        elif a.dtype == np.int64:
            dt = I2V_I64
        elif a.dtype == np.uint8:
            dt = I2V_U8
        else:
            raise I2VError(f"state_dict entry {k}: unsupported dtype {a.dtype}")
        kb = k.encode()
        keep.append((kb, a))
        items.append(_Tensor(kb, a.ctypes.data_as(c_void_p), a.size, dt))
    arr = (_Tensor * len(items))(*items)
    return arr, keep


class _Workspace:
    """Caller-owned device scratch, cached per (device, size class) so its address is stable across calls."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


def _on_device(fn):
    """Method decorator of the native handles: reject tensors that live on another GPU, run with the handle's device current."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        tensors = [t for t in list(args) + list(kwargs.values()) if isinstance(t, torch.Tensor)]
        with self._on(*tensors):
            return fn(self, *args, **kwargs)
    return wrapper


class NativeFlow(_Handle):
    """Handle for ``i2v_flow_*`` (ConditionalFlow, flow_blocks.py:8-60)."""

    def __init__(self, in_channels, embedding_dim, hidden_dim, hidden_depth, n_flows, control=False,
                 activation="lrelu", skip_actnorm=False, skip_shuffle=False, use_graph=True, device=None, linear_f16=None):
        self.linear_f16 = default_flow_f16() if linear_f16 is None else int(bool(linear_f16))
        cfg = FlowCfg(in_channels, embedding_dim, hidden_dim, hidden_depth, n_flows, int(control),
                      1 if activation == "lrelu" else 0, int(skip_actnorm), int(skip_shuffle), int(use_graph), self.linear_f16)
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_flow_create(ctypes.byref(cfg), ctypes.byref(h)), "i2v_flow_create")
        self._h = h
        self.embedding_dim = embedding_dim
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_flow_destroy(self._h)
            self._h = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_flow_load(self._h, arr, len(arr)), "i2v_flow_load")
        del keep

    @property
    def param_bytes(self):
        return int(lib().i2v_flow_param_bytes(self._h))

    @_on_device
    def _run(self, x, embed, reverse):
        _require_gpu(x, embed)
        B = x.shape[0]
        if x.shape != (B, 64) or embed.shape != (B, self.embedding_dim):
            raise I2VError(f"flow: expected x [B,64] and embed [B,{self.embedding_dim}], got {tuple(x.shape)}, {tuple(embed.shape)}")
        nbytes = lib().i2v_flow_workspace_bytes(self._h, B)
        ws = self._ws.get(nbytes, x.device)
        out = torch.empty_like(x)
        if reverse:
            _check(lib().i2v_flow_inverse(self._h, x.data_ptr(), embed.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                          ws.numel(), B, _stream()), "i2v_flow_inverse")
            return out
        logdet = torch.empty(B, dtype=torch.float32, device=x.device)
        _check(lib().i2v_flow_forward(self._h, x.data_ptr(), embed.data_ptr(), out.data_ptr(), logdet.data_ptr(),
                                      ws.data_ptr(), ws.numel(), B, _stream()), "i2v_flow_forward")
        return out, logdet

    @_on_device
    def forward(self, x, embed):
        return self._run(x, embed, False)

    def inverse(self, x, embed):
        return self._run(x, embed, True)


class NativeDecoder(_Handle):
    """Handle for ``i2v_dec_*`` (Generator, decoder.py:55-120)."""

    def __init__(self, channel_factor, z_dim, upsample_s, upsample_t, spectral_norm=True, mma=0, device=None):
        cfg = DecCfg(channel_factor, z_dim, (c_int32 * 2)(*upsample_s), (c_int32 * 2)(*upsample_t),
                     int(bool(spectral_norm)), mma)
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_dec_create(ctypes.byref(cfg), ctypes.byref(h)), "i2v_dec_create")
        self._h = h
        self.z_dim = z_dim
        self._ws = _Workspace()
        t, hh, w = c_int32(), c_int32(), c_int32()
        _check(lib().i2v_dec_out_shape(self._h, ctypes.byref(t), ctypes.byref(hh), ctypes.byref(w)), "i2v_dec_out_shape")
        self.out_shape = (t.value, hh.value, w.value)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_dec_destroy(self._h)     # (synchronises the side stream; a shared one is still alive: _side_stream is released below)
            self._h = None
        self._side_stream = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_dec_load(self._h, arr, len(arr)), "i2v_dec_load")
        del keep

    def flops_per_sample(self, img_h, img_w):
        return float(lib().i2v_dec_flops_per_sample(self._h, img_h, img_w))

    @_on_device
    def set_profile(self, on):
        _check(lib().i2v_dec_set_profile(self._h, int(on)), "i2v_dec_set_profile")

    @_on_device
    def get_profile(self):
        a, b, e, c = c_double(), c_double(), c_double(), c_int64()
        _check(lib().i2v_dec_get_profile(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(e), ctypes.byref(c)),
               "i2v_dec_get_profile")
        return {"conv3_ms": a.value, "conv3_flops": b.value, "conv3_mfma_flops": e.value, "conv3_launches": c.value}

    @_on_device
    def get_layer_profile(self):
        """Per-layer totals of the profiled 3x3x3 conv launches since set_profile(True) (i2v_dec_get_layer_profile)."""
        rows = []
        for layer in range(12):
            name = ctypes.create_string_buffer(48)
            ms, fl, ex, n, k = c_double(), c_double(), c_double(), c_int64(), c_int32()
            _check(lib().i2v_dec_get_layer_profile(self._h, layer, name, 48, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ex),
                                                   ctypes.byref(n), ctypes.byref(k)), "i2v_dec_get_layer_profile")
            if n.value:
                rows.append({"layer": name.value.decode(), "kernel": ("conv_mfma_f32", "conv_mfma_f16x3", "conv_wino_f16x3", "conv_wino4_f16x3", "conv_wino4g_f16x3")[k.value],
                             "launches": int(n.value), "ms": ms.value, "flops": fl.value, "mfma_flops": ex.value})
        return rows

    @_on_device
    def status(self, reset=False):
        """Range guard of the split-fp16 operand format (i2v_dec_status): synchronises the current stream and returns the
        sticky flag word (bit 0: an activation left the fp16 range, the outputs since then are invalid)."""
        flags = c_int32()
        _check(lib().i2v_dec_status(self._h, ctypes.byref(flags), int(bool(reset)), _stream()), "i2v_dec_status")
        return int(flags.value)

    LAYER_NAMES = tuple(f"{b}.conv_{i}" for b in ("head_0", "g_0", "g_1", "g_2", "g_3", "g_4") for i in (0, 1))

    def fallback_layers(self):
        """mma = auto (i2v_dec_fallback_layers): the 3x3x3 convs the range guard has switched to the exact-fp32 kernels so far, as
        ``{"layers": [names], "whole_handle": bool, "reruns": n}``.  Empty for a checkpoint inside the split format's window."""
        mask, reruns = c_int32(), c_int32()
        _check(lib().i2v_dec_fallback_layers(self._h, ctypes.byref(mask), ctypes.byref(reruns)), "i2v_dec_fallback_layers")
        return {"layers": [n for i, n in enumerate(self.LAYER_NAMES) if mask.value >> i & 1], "whole_handle": bool(mask.value >> 30 & 1),
                "reruns": int(reruns.value)}

    @_on_device
    def debug_tap(self, block, which, dst):
        """Test hook (i2v_dec_debug_tap): dst = float32 CUDA tensor or None."""
        if dst is None:
            _check(lib().i2v_dec_debug_tap(self._h, -1, -1, None, 0), "i2v_dec_debug_tap")
        else:
            _check(lib().i2v_dec_debug_tap(self._h, block, which, dst.data_ptr(), dst.numel()), "i2v_dec_debug_tap")

    @_on_device
    def set_side_stream(self, stream):
        """i2v_dec_set_side_stream: run the handle's side work (SPADE branches, learned shortcuts, ``prepare``) on ``stream`` (a
        ``torch.cuda.Stream``, e.g. ``LatentPrefetcher.stream``) instead of a stream of the handle's own; ``None`` restores that.
        The binding keeps the stream object alive for as long as the handle uses it."""
        if stream is not None and stream.device != self.device:
            raise I2VError(f"side stream on {stream.device} for a handle on {self.device}")
        _check(lib().i2v_dec_set_side_stream(self._h, c_void_p(stream.cuda_stream) if stream is not None else None),
               "i2v_dec_set_side_stream")
        self._side_stream = stream
        self._prep = None

    def _workspace(self, nbytes, device):
        """The handle's workspace; before it is REPLACED by a larger one, the current stream joins the handle's side stream (a forked
        prepare may still be writing the old buffer: i2v_dec_join), so the caching allocator's stream-ordered free is safe."""
        buf = self._ws.buf
        if buf is not None and (buf.numel() < nbytes or buf.device != device):
            _check(lib().i2v_dec_join(self._h, _stream()), "i2v_dec_join")
            self._prep = None
        return self._ws.get(nbytes, device)

    @staticmethod
    def _version(t):
        """Version counter of a tensor, None for inference tensors (they do not track one: RuntimeError on access)."""
        try:
            return t._version
        except RuntimeError:
            return None

    @_on_device
    def prepare(self, img):
        """i2v_dec_prepare: enqueue the SPADE branches of all six blocks (they depend on the start frame only) on the HANDLE's side
        stream, ordered behind everything already on the current stream; the next ``forward`` with the SAME tensor (same storage,
        batch, size) waits for them per level instead of computing them.  The current stream stays free (e.g. for the cINN pass).
        Inference tensors (``torch.inference_mode``) carry no version counter, so an in-place refill between the prepare and its
        forward could not be detected: for them this is a no-op and the forward computes the branches itself (same bits)."""
        # every check BEFORE any state changes: a raise must leave the Python side and the C side agreeing (nothing prepared)
        if getattr(self, "_prep", None) is not None:
            self._prep = None
            _check(lib().i2v_dec_prepare_cancel(self._h), "i2v_dec_prepare_cancel")
        _require_gpu(img)
        B = img.shape[0]
        if img.dim() != 4 or img.shape[1] != 3:
            raise I2VError(f"decoder: expected img [B,3,H,W], got {tuple(img.shape)}")
        ver = self._version(img)
        if ver is None:
            return
        nbytes = lib().i2v_dec_workspace_bytes(self._h, B, img.shape[2], img.shape[3])
        ws = self._workspace(nbytes, img.device)
        _check(lib().i2v_dec_prepare(self._h, img.data_ptr(), img.shape[2], img.shape[3], ws.data_ptr(), ws.numel(), B, _stream()),
               "i2v_dec_prepare")
        # the C side recognises the prepared frames by ADDRESS; a caching allocator hands the same address to the next same-size
        # tensor and a buffer refilled in place keeps it, so the binding also remembers WHICH tensor (weak) and its version
        self._prep = (weakref.ref(img), ver)

    @staticmethod
    def _sample_strided(t, inner_shape):
        """(data_ptr, sample stride in floats) of a float32 tensor [B, *inner_shape] whose samples are contiguous blocks."""
        if t.dtype != torch.float32 or tuple(t.shape[1:]) != tuple(inner_shape):
            return None
        inner = 1
        for n, s in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
            if n != 1 and s != inner:
                return None
            inner *= n
        bs = t.stride(0) if t.shape[0] > 1 else inner
        return (t.data_ptr(), bs) if bs >= inner else None

    @_on_device
    def forward(self, img, motion, out=None):
        """i2v_dec_forward_strided.  ``img``: [B,3,H,W] whose samples are contiguous [3,H,W] blocks (any sample stride: e.g. the view
        ``seq[:, -1]`` of a [B,T,3,H,W] buffer).  ``out``: optional float32 view [B,T,3,H,W] with contiguous [T,3,H,W] sample blocks
        (e.g. ``buf[:, 16:32]`` of a [B,32,3,H,W] buffer) that receives the frames; a dense tensor is allocated otherwise."""
        for t in (img, motion):
            if not t.is_cuda:
                _require_gpu(t)   # raises: no CPU fallback
            if t.dtype != torch.float32:
                raise I2VError(f"expected float32 tensors, got {t.dtype}")
        prep, self._prep = getattr(self, "_prep", None), None
        if prep is not None and (prep[0]() is not img or prep[1] != self._version(img)):
            _check(lib().i2v_dec_prepare_cancel(self._h), "i2v_dec_prepare_cancel")   # another tensor, or this one was written since
        B = img.shape[0]
        if img.dim() != 4 or img.shape[1] != 3 or motion.shape != (B, self.z_dim):
            raise I2VError(f"decoder: expected img [B,3,H,W] and motion [B,{self.z_dim}], got {tuple(img.shape)}, {tuple(motion.shape)}")
        iv = self._sample_strided(img, img.shape[1:])
        if iv is None:
            img = img.contiguous()
            iv = (img.data_ptr(), 3 * img.shape[2] * img.shape[3])
        if not motion.is_contiguous():
            motion = motion.contiguous()
        nbytes = lib().i2v_dec_workspace_bytes(self._h, B, img.shape[2], img.shape[3])
        ws = self._workspace(nbytes, img.device)
        T, H, W = self.out_shape
        if out is None:
            out = torch.empty(B, T, 3, H, W, dtype=torch.float32, device=img.device)
        ov = self._sample_strided(out, (T, 3, H, W)) if (out.is_cuda and out.device == img.device and out.shape[0] == B) else None
        if ov is None:
            raise I2VError(f"decoder: out must be a float32 [B={B},{T},3,{H},{W}] view on {img.device} with contiguous sample blocks, "
                           f"got {tuple(out.shape)} strides {out.stride()}")
        _check(lib().i2v_dec_forward_strided(self._h, iv[0], img.shape[2], img.shape[3], iv[1], motion.data_ptr(), ov[0], ov[1],
                                             ws.data_ptr(), ws.numel(), B, _stream()), "i2v_dec_forward_strided")
        return out


class NativeMLP(_Handle):
    """Handle for ``i2v_mlp_*`` (BasicFullyConnectedNet, modules.py:9-30)."""

    def __init__(self, dim, hidden_dim, depth, out_dim, device=None):
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_mlp_create(dim, hidden_dim, depth, out_dim, ctypes.byref(h)), "i2v_mlp_create")
        self._h = h
        self.dim, self.out_dim = dim, out_dim
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_mlp_destroy(self._h)
            self._h = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_mlp_load(self._h, arr, len(arr)), "i2v_mlp_load")
        del keep

    @_on_device
    def forward(self, x):
        _require_gpu(x)
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise I2VError(f"mlp: expected x [B,{self.dim}], got {tuple(x.shape)}")
        B = x.shape[0]
        ws = self._ws.get(lib().i2v_mlp_workspace_bytes(self._h, B), x.device)
        y = torch.empty(B, self.out_dim, dtype=torch.float32, device=x.device)
        _check(lib().i2v_mlp_forward(self._h, x.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), B, _stream()),
               "i2v_mlp_forward")
        return y


OP_ACTNORM_FWD, OP_ACTNORM_REV, OP_INVLRELU_FWD, OP_INVLRELU_REV, OP_GATHER = range(5)


def _channel_op(op, x, p0=None, p1=None, idx=None, alpha=0.0):
    _require_gpu(x)
    if x.dim() < 2:
        raise I2VError("channel op: need a [B, C, ...] tensor")
    B, C = x.shape[0], x.shape[1]
    inner = x.numel() // (B * C)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().i2v_channel_op(op, x.data_ptr(), out.data_ptr(), B, C, inner,
                                    p0.data_ptr() if p0 is not None else None, p1.data_ptr() if p1 is not None else None,
                                    idx.data_ptr() if idx is not None else None, float(alpha), _stream()), "i2v_channel_op")
    return out


def actnorm(x, loc, scale, reverse):
    """ActNorm.forward / reverse arithmetic (modules.py:80,100) on [B,C,H,W]."""
    loc = loc.detach().reshape(-1).contiguous()
    scale = scale.detach().reshape(-1).contiguous()
    _require_gpu(loc, scale)
    return _channel_op(OP_ACTNORM_REV if reverse else OP_ACTNORM_FWD, x, loc, scale)


def actnorm_logdet(scale, hw, batch):
    scale = scale.detach().reshape(-1).contiguous()
    _require_gpu(scale)
    out = torch.empty(batch, dtype=torch.float32, device=scale.device)
    with torch.cuda.device(scale.device):
        _check(lib().i2v_actnorm_logdet(scale.data_ptr(), scale.numel(), float(hw), out.data_ptr(), batch, _stream()),
               "i2v_actnorm_logdet")
    return out


def probe_mfma_f16(device, workgroups=2048, iters=4096, reps=3):
    """Sustained fp16 matrix-core rate with live operands (TFLOP/s of v_mfma_f32_32x32x16_f16 actually executed by an
    MFMA-only loop of the conv kernel's shape; measurement helper for bench.py).  Median of ``reps`` event-timed launches
    after one warm-up launch of the same length (so the clock has settled)."""
    scratch = torch.empty(workgroups * 512, dtype=torch.float32, device=device)
    flops = c_double()
    rates = []
    with torch.cuda.device(scratch.device):
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _check(lib().i2v_probe_mfma_f16(workgroups, iters, scratch.data_ptr(), ctypes.byref(flops), _stream()),
                   "i2v_probe_mfma_f16")
            e1.record()
            e1.synchronize()
            if r:
                rates.append(flops.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    rates.sort()
    return rates[len(rates) // 2]


def inv_lrelu(x, alpha, reverse):
    return _channel_op(OP_INVLRELU_REV if reverse else OP_INVLRELU_FWD, x, alpha=alpha)


def gather_channels(x, idx):
    idx = idx.detach().contiguous()
    if not idx.is_cuda or idx.dtype != torch.int64:
        raise I2VError("gather_channels: idx must be an int64 tensor on the GPU")
    return _channel_op(OP_GATHER, x, idx=idx)


def channel_mean_std(flat):
    """flat [C, N] -> (mean [C], unbiased std [C])."""
    _require_gpu(flat)
    C, N = flat.shape
    mean = torch.empty(C, dtype=torch.float32, device=flat.device)
    std = torch.empty(C, dtype=torch.float32, device=flat.device)
    with torch.cuda.device(flat.device):
        _check(lib().i2v_row_mean_std(flat.data_ptr(), C, N, mean.data_ptr(), std.data_ptr(), _stream()), "i2v_row_mean_std")
    return mean, std


def default_flow_f16():
    """Operand precision of the cINN's Linear layers: 0 = exact fp32 matrix cores (default), 1 = fp16 operands with fp32
    accumulation (BASELINE configs[4]; i2v_flow_cfg.linear_f16); env I2V_FLOW_F16."""
    return int(os.environ.get("I2V_FLOW_F16", "0"))


def parse_mma(v):
    """0 / 1 / 2 or "auto" (= 2) -> the i2v_dec_cfg.mma value."""
    if isinstance(v, str):
        v = v.strip().lower()
        return 2 if v == "auto" else int(v)
    return int(v)


def default_mma():
    """Matrix-core mode of the 3x3x3 convolutions: 1 = split-fp16 (default), 0 = exact fp32 MFMA, 2 / "auto" = split-fp16 with the
    per-layer fallback to exact fp32 behind the range guard (every forward synchronises); env I2V_DEC_MMA."""
    return parse_mma(os.environ.get("I2V_DEC_MMA", "1"))


class NativeGBlock(_Handle):
    """Handle for ``i2v_gblock_*`` (GeneratorBlock, decoder.py:7-52; tensors in the reference layout [B,C,T,H,W])."""

    def __init__(self, n_in, n_out, z_dim, spectral_norm=True, mma=None, device=None):
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_gblock_create(n_in, n_out, z_dim, int(bool(spectral_norm)), default_mma() if mma is None else mma,
                                           ctypes.byref(h)), "i2v_gblock_create")
        self._h = h
        self.n_in, self.n_out, self.n_mid, self.z_dim = n_in, n_out, min(n_in, n_out), z_dim
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_gblock_destroy(self._h)
            self._h = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_gblock_load(self._h, arr, len(arr)), "i2v_gblock_load")
        del keep

    def _geom(self, x, channels):
        if x.dim() != 5 or x.shape[1] != channels:
            raise I2VError(f"expected x [B,{channels},T,H,W], got {tuple(x.shape)}")
        B, _, T, H, W = x.shape
        ws = self._ws.get(lib().i2v_gblock_workspace_bytes(self._h, B, T, H, W), x.device)
        return B, T, H, W, ws

    @_on_device
    def forward(self, x, z, img):
        _require_gpu(x, z, img)
        B, T, H, W, ws = self._geom(x, self.n_in)
        if z.shape != (B, self.z_dim) or img.dim() != 4 or img.shape[:2] != (B, 3):
            raise I2VError(f"GeneratorBlock: expected z [B,{self.z_dim}] and img [B,3,h,w], got {tuple(z.shape)}, {tuple(img.shape)}")
        out = torch.empty(B, self.n_out, T, H, W, dtype=torch.float32, device=x.device)
        _check(lib().i2v_gblock_forward(self._h, x.data_ptr(), z.data_ptr(), img.data_ptr(), img.shape[2], img.shape[3],
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, W, _stream()), "i2v_gblock_forward")
        return out

    @_on_device
    def status(self, reset=False):
        """Range guard of this block's split-fp16 operand writers (i2v_gblock_status): synchronises, returns the flag word."""
        flags = c_int32()
        _check(lib().i2v_gblock_status(self._h, ctypes.byref(flags), int(bool(reset)), _stream()), "i2v_gblock_status")
        return int(flags.value)

    @_on_device
    def norm(self, part, x, cond):
        _require_gpu(x, cond)
        B, T, H, W, ws = self._geom(x, self.n_mid if part == 1 else self.n_in)
        ih = iw = 0
        if part == 0:
            if cond is None or cond.dim() != 4 or cond.shape[:2] != (B, 3):
                raise I2VError("Spade: expected the start frame [B,3,h,w]")
            ih, iw = cond.shape[2], cond.shape[3]
        if part == 1 and (cond is None or cond.shape != (B, self.z_dim)):
            raise I2VError(f"ADAIN: expected z [B,{self.z_dim}]")
        out = torch.empty_like(x)
        _check(lib().i2v_gblock_norm(self._h, part, x.data_ptr(), cond.data_ptr() if cond is not None else None, ih, iw,
                                     out.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, W, _stream()), "i2v_gblock_norm")
        return out


class NativeNorm:
    """A lone Spade / ADAIN / Norm3D (normalization_layer.py:5-51) on top of a partially loaded ``i2v_gblock``."""
    _PART = {"spade": (0, "norm_0."), "adain": (1, "norm_1."), "norm3d": (2, "norm_s.")}

    def __init__(self, kind, num_features, z_dim, mma=None, device=None):
        self.part, self.prefix = self._PART[kind]
        self.blk = NativeGBlock(num_features, num_features, z_dim if z_dim else 64, spectral_norm=False, mma=mma, device=device)

    def load(self, state_dict):
        self.blk.load({self.prefix + k: v for k, v in state_dict.items()})

    def forward(self, x, cond):
        return self.blk.norm(self.part, x, cond)


class NativeEmbedder(_Handle):
    """Handle for ``i2v_embedder_*`` (ResnetEncoder.encode(x).mode(), AE.py:91-166)."""

    def __init__(self, z_dim, use_batchnorm, device=None):
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_embedder_create(z_dim, int(bool(use_batchnorm)), ctypes.byref(h)), "i2v_embedder_create")
        self._h = h
        self.z_dim = z_dim
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_embedder_destroy(self._h)
            self._h = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_embedder_load(self._h, arr, len(arr)), "i2v_embedder_load")
        del keep

    @_on_device
    def forward(self, img):
        _require_gpu(img)
        if img.dim() != 4 or img.shape[1] != 3:
            raise I2VError(f"embedder: expected img [B,3,H,W], got {tuple(img.shape)}")
        B, _, H, W = img.shape
        ws = self._ws.get(lib().i2v_embedder_workspace_bytes(self._h, B, H, W), img.device)
        out = torch.empty(B, self.z_dim, dtype=torch.float32, device=img.device)
        _check(lib().i2v_embedder_forward(self._h, img.data_ptr(), H, W, out.data_ptr(), ws.data_ptr(), ws.numel(), B, _stream()),
               "i2v_embedder_forward")
        return out


class NativeEncoder3D(_Handle):
    """Handle for ``i2v_encoder3d_*`` (Encoder.forward, resnet3D.py:138-219)."""

    def __init__(self, z_dim, channels, stride_s, stride_t, device=None):
        cfg = Enc3dCfg(z_dim, (c_int32 * 5)(*channels), (c_int32 * 4)(*stride_s), (c_int32 * 4)(*stride_t), 0)
        h = c_void_p()
        with self._bind(device):
            _check(lib().i2v_encoder3d_create(ctypes.byref(cfg), ctypes.byref(h)), "i2v_encoder3d_create")
        self._h = h
        self.z_dim = z_dim
        self._ws = _Workspace()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.i2v_encoder3d_destroy(self._h)
            self._h = None

    @_on_device
    def load(self, state_dict):
        arr, keep = _pack_state_dict(state_dict)
        _check(lib().i2v_encoder3d_load(self._h, arr, len(arr)), "i2v_encoder3d_load")
        del keep

    @_on_device
    def forward(self, x, eps=None):
        _require_gpu(x, eps)
        if x.dim() != 5 or x.shape[1] != 3:
            raise I2VError(f"encoder: expected x [B,3,T,H,W], got {tuple(x.shape)}")
        B, _, T, H, W = x.shape
        ws = self._ws.get(lib().i2v_encoder3d_workspace_bytes(self._h, B, T, H, W), x.device)
        mu = torch.empty(B, self.z_dim, dtype=torch.float32, device=x.device)
        logvar = torch.empty_like(mu)
        sample = torch.empty_like(mu) if eps is not None else None
        _check(lib().i2v_encoder3d_forward(self._h, x.data_ptr(), T, H, W, eps.data_ptr() if eps is not None else None,
                                           sample.data_ptr() if sample is not None else None, mu.data_ptr(), logvar.data_ptr(),
                                           ws.data_ptr(), ws.numel(), B, _stream()), "i2v_encoder3d_forward")
        return sample, mu, logvar
