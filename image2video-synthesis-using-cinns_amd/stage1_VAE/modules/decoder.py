"""Mirror of the reference's ``stage1_VAE/modules/decoder.py`` class surface (GeneratorBlock, Generator).

``Generator.forward(img, motion)`` is ONE call into libi2v_hip.so (``i2v_dec_forward``, csrc/i2v_dec.hip): all
activations stay channels-last in HBM, spectral-norm is folded at load time, the 3x3x3 convolutions run on the
gfx950 matrix cores.  The sub-modules carry the reference's state_dict keys (``g_k.conv_0.weight_orig`` ...) so
released ``.pth`` files load unchanged."""
import torch
import torch.nn as nn

import i2v_native as native
from i2v_params import ConvParams, LinearParams, NativeBacked
from stage1_VAE.modules.normalization_layer import ADAIN, Norm3D, Spade


class GeneratorBlock(NativeBacked):
    """out = shortcut(x) + conv_1(lrelu(ADAIN(conv_0(lrelu(Spade(x, img))), z)))   (reference decoder.py:7-52)."""

    def __init__(self, n_in, n_out, use_spectral, z_dim):
        super().__init__()
        self.learned_shortcut = (n_in != n_out)
        n_middle = min(n_in, n_out)
        self.n_in, self.n_out, self.z_dim, self.use_spectral = n_in, n_out, z_dim, bool(use_spectral)
        self.conv_0 = ConvParams(n_in, n_middle, 3, 3, bias=True, spectral=use_spectral)
        self.conv_1 = ConvParams(n_middle, n_out, 3, 3, bias=True, spectral=use_spectral)
        if self.learned_shortcut:
            self.conv_s = ConvParams(n_in, n_out, 1, 3, bias=False, spectral=use_spectral)
        self.norm_0 = Spade(n_in)
        self.norm_1 = ADAIN(n_middle, z_dim)
        if self.learned_shortcut:
            self.norm_s = Norm3D(n_in)

    def _build_native(self):
        h = native.NativeGBlock(self.n_in, self.n_out, self.z_dim, self.use_spectral, device=self.module_device())
        h.load(self.state_dict())
        return h

    def forward(self, x, cond1, cond2):
        return self.native().forward(x.contiguous(), cond1.contiguous(), cond2.contiguous())


class Generator(NativeBacked):
    """fc -> head_0 -> (x2, g_0) -> (x2, g_1) -> (x2, g_2) -> (up, g_3) -> (up, g_4) -> lrelu -> conv_img -> tanh
    (reference decoder.py:55-120).  Returns ``[B, 16, 3, H, W]`` (contiguous; the reference returns a transposed
    view of a [B,3,16,H,W] buffer)."""

    def __init__(self, dic):
        super().__init__()
        nf = dic["channel_factor"]
        self.z_dim = dic["z_dim"]
        self.fmap_start = 16 * nf
        use_spectral = dic["spectral_norm"]
        self.upsample_s = list(dic["upsample_s"])
        self.upsample_t = list(dic["upsample_t"])
        self.channel_factor = nf
        self.use_spectral = bool(use_spectral)
        self.fc = LinearParams(self.z_dim, 4 * 4 * 16 * nf)
        self.head_0 = GeneratorBlock(16 * nf, 16 * nf, use_spectral, self.z_dim)
        self.g_0 = GeneratorBlock(16 * nf, 16 * nf, use_spectral, self.z_dim)
        self.g_1 = GeneratorBlock(16 * nf, 8 * nf, use_spectral, self.z_dim)
        self.g_2 = GeneratorBlock(8 * nf, 4 * nf, use_spectral, self.z_dim)
        self.g_3 = GeneratorBlock(4 * nf, 2 * nf, use_spectral, self.z_dim)
        self.g_4 = GeneratorBlock(2 * nf, 1 * nf, use_spectral, self.z_dim)
        self.conv_img = ConvParams(nf, 3, 3, 3, bias=True, spectral=False)
        # matrix-core mode of the 3x3x3 convolutions: 0 = exact fp32 MFMA, 1 = split-fp16 (3 fp16 MFMAs per product,
        # fp32-class accuracy, ~5x the rate).  Not a reference key: taken from dic["mma"] or the I2V_DEC_MMA env var.
        mma = dic.get("mma", None) if hasattr(dic, "get") else None
        self.mma = native.default_mma() if mma is None else native.parse_mma(mma)   # 0, 1, 2 or "auto" (= 2)

    def _build_native(self):
        h = native.NativeDecoder(self.channel_factor, self.z_dim, self.upsample_s, self.upsample_t, self.use_spectral,
                                 mma=self.mma, device=self.module_device())
        h.load(self.state_dict())
        if getattr(self, "_shared_side", None) is not None and self._shared_side.device == h.device:
            h.set_side_stream(self._shared_side)
        return h

    def forward(self, img, motion, out=None):
        """decoder.py:97-120.  ``out`` (not a reference argument): a float32 view [B,16,3,H,W] with contiguous sample blocks to
        decode into; ``img`` may be a sample-strided view such as ``seq[:, -1]`` -- together they let the autoregressive loop of
        ``Model.forward`` (get_model.py:68-73) run without ``torch.cat`` / ``.contiguous()`` copies."""
        return self.native().forward(img, motion, out=out)

    def decode_sequence(self, x_0, z, vid_length):
        """get_model.py:68-73 -- ``seq = G(x_0, z); while T < vid_length: seq = cat(seq, G(seq[:, -1], z))`` -- into ONE
        pre-allocated [B, 16*ceil(vid_length/16), 3, H, W] buffer: pass k writes frames [16k, 16k+16) in place and pass k+1 reads
        its start frames from the strided view seq[:, 16k+15] (i2v_dec_forward_strided).  Same kernels, same bits as the loop."""
        T, H, W = self.native().out_shape
        n = max(1, -(-int(vid_length) // T))
        if n == 1:
            return self.forward(x_0, z)
        seq = torch.empty(x_0.shape[0], n * T, 3, H, W, dtype=torch.float32, device=x_0.device)
        self.forward(x_0, z, out=seq[:, :T])
        for k in range(1, n):
            self.forward(seq[:, k * T - 1], z, out=seq[:, k * T:(k + 1) * T])
        return seq

    def share_side_stream(self, stream):
        """Not a reference method: run the decoder's side work (SPADE branches, learned shortcuts, ``prepare``) on ``stream`` -- the
        stream the caller's cINN prefetch runs on (``i2v_pipeline.LatentPrefetcher.stream``) -- instead of a stream the native handle
        creates: one side stream per job (``None``: the handle's own again).  Survives a rebuild of the native handle."""
        object.__setattr__(self, "_shared_side", stream)
        self.native().set_side_stream(stream)

    def prepare(self, img):
        """Not a reference method: enqueue the SPADE branches of all blocks for the start frames ``img`` (they do not depend on
        the motion latent) on the native handle's side stream, behind what is already on the current stream.  The next
        ``forward(img, z)`` with the SAME contiguous tensor waits for them level by level instead of computing them
        (i2v_dec_prepare); ``get_model.Model.synthesize`` uses it to fill the time the cINN pass takes."""
        self.native().prepare(img.contiguous())
