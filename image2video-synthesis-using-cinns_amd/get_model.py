"""Inference facade, mirror of the reference's ``get_model.py``: ``Model(model_path, vid_length)`` builds the
decoder and the cINN from two YAML files and their checkpoints; ``forward(x_0)`` = sample residual -> cINN inverse
-> decode (-> autoregressive repeats).

Additions over the reference surface (all optional, defaults reproduce the reference):
  * ``forward(x_0, cond=None, residual=None, embed=None)``: the latent draw and the conditioning embedding can be
    supplied so that "identical latent samples" is testable;
  * ``synthesize(...)``: the same computation WITHOUT the final ``seq[:vid_length]`` slice, which in the reference
    slices the BATCH dimension (quirk Q3, get_model.py:75); the benchmark and the multi-GPU harness call this."""
import os

import torch

import i2v_config
from stage1_VAE.modules import decoder
from stage1_VAE.modules.resnet3D import Encoder
from stage2_cINN.modules import INN


class Model(torch.nn.Module):
    def __init__(self, model_path, vid_length, transfer=False, embedder=None):
        super().__init__()
        opt = i2v_config.load(os.path.join(model_path, "config_stage2.yaml"))
        path_stage1 = opt.First_stage_model["model_path"] + opt.First_stage_model["model_name"] + "/"
        config = i2v_config.load(path_stage1 + "config_stage1.yaml")

        # Matrix-core mode: the checkpoint's activations are unknown until they have been seen, so the drop-in entry point runs the
        # decoder in AUTO mode (split-fp16 with the per-layer fallback to the exact-fp32 kernels behind the range guard: a checkpoint
        # inside the split format's window runs exactly the default launches, one outside it still returns valid frames -- check()
        # says which layers were switched) unless the YAML's Decoder section or I2V_DEC_MMA pick a mode.
        dec_cfg = dict(config.Decoder)
        if "mma" not in dec_cfg and "I2V_DEC_MMA" not in os.environ:
            dec_cfg["mma"] = "auto"
        self.decoder = decoder.Generator(dec_cfg).cuda()
        self.decoder.load_state_dict(torch.load(path_stage1 + opt.First_stage_model["checkpoint_decoder"] + ".pth",
                                                map_location="cpu")["state_dict"])
        _ = self.decoder.eval()

        if transfer:   # motion encoder of the transfer path (get_model.py:27-31)
            self.encoder = Encoder(dic=config.Encoder).cuda()
            self.encoder.load_state_dict(torch.load(path_stage1 + opt.First_stage_model["checkpoint_encoder"] + ".pth.tar",
                                                    map_location="cpu")["state_dict"])
            _ = self.encoder.eval()

        flow_mid_channels = config.Decoder["z_dim"] * opt.Flow["flow_mid_channels_factor"]
        self.flow = INN.SupervisedTransformer(flow_in_channels=config.Decoder["z_dim"],
                                              flow_embedding_channels=opt.Conditioning_Model["z_dim"],
                                              n_flows=opt.Flow["n_flows"],
                                              flow_hidden_depth=opt.Flow["flow_hidden_depth"],
                                              flow_mid_channels=flow_mid_channels,
                                              flow_conditioning_option="None",
                                              dic=opt.Conditioning_Model,
                                              control=bool(opt.Training["control"]),
                                              embedder=embedder).cuda()
        self.flow.flow.load_state_dict(torch.load(os.path.join(model_path, "cINN.pth"), map_location="cpu")["state_dict"])
        _ = self.flow.eval()

        self.z_dim = config.Decoder["z_dim"]
        self.vid_length = vid_length
        self.config = opt
        self.overlap = True       # synthesize(): cINN pass on a side stream underneath the decoder's SPADE branches
        self._prefetch = None

    @torch.no_grad()
    def sample_latent(self, x_0, cond=None, residual=None, embed=None):
        """First half of ``forward`` (get_model.py:59-66): draw the residual, cINN inverse -> z [B, z_dim]."""
        if residual is None:
            residual = torch.randn(x_0.size(0), self.z_dim).cuda()  # CPU generator, like get_model.py:59
        return self.flow(residual, [x_0, cond], reverse=True, embed=embed).view(x_0.size(0), -1)

    @torch.no_grad()
    def decode(self, x_0, z):
        """Second half (get_model.py:68-73): decoder pass(es), autoregressive on the last frame with the same z."""
        return self.decoder.decode_sequence(x_0, z, self.vid_length)   # the same loop, decoded in place into one buffer

    @torch.no_grad()
    def synthesize(self, x_0, cond=None, residual=None, embed=None):
        """[B,3,H,W] -> [B, 16*ceil(vid_length/16), 3, H, W]; no batch slice.
        ONE call overlaps its own two halves where they are independent: the cINN pass (an 82-launch dependent chain that leaves
        most of the chip idle) runs on a high-priority side stream, the decoder's SPADE branches -- they depend on the start frame
        only -- on the decoder handle's own side stream (``Generator.prepare``), and the decoder proper starts as soon as the latent
        is there, its blocks waiting for their level's maps (``overlap = False`` restores the strictly serial order; the frames are
        the same bits either way)."""
        # (single-stream semantics are kept while the caller captures a graph: a side stream cannot be forked inside a capture here)
        if not (self.overlap and x_0.is_cuda) or torch.cuda.is_current_stream_capturing():
            return self.decode(x_0, self.sample_latent(x_0, cond, residual, embed))
        import i2v_pipeline
        if self._prefetch is None or self._prefetch.stream.device != x_0.device:
            self._prefetch = i2v_pipeline.LatentPrefetcher(lambda a, b, c, d: self.sample_latent(a, b, c, d), device=x_0.device)
            # Three streams at every N: the caller's, this prefetch stream, the decoder handle's own side stream.  A rank of a multi-GPU
            # job that collates a STREAM of calls off the compute stream issues its all-gathers on the prefetch stream too
            # (``self.collator(total)``), not on a fourth stream: HIP multiplexes streams onto four hardware queues, and a collation
            # stream of its own cost +7 % (B = 64) / +46 % (B = 8) per step, the shared-side-stream remedy +2 % / +9 %, the gathers on
            # the prefetch stream nothing (profiles/r06_c_stream_configurations.txt, r06_o_collation_on_prefetch_stream.txt).
            self.decoder.share_side_stream(None)
        x_0 = x_0.contiguous()
        ticket = self._prefetch.submit(x_0, cond, residual, embed)
        self.decoder.prepare(x_0)
        return self.decode(x_0, self._prefetch.get(ticket))

    def collator(self, total, group=None, device=None):
        """``i2v_dist.OverlappedCollator`` for a stream of ``synthesize`` calls on this rank's shard of a ``total``-sample job: every
        call's all-gather is issued on the stream the cINN prefetch runs on (see ``synthesize``), overlapping the next call."""
        import i2v_dist
        import i2v_pipeline
        if self._prefetch is None:
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            self._prefetch = i2v_pipeline.LatentPrefetcher(lambda a, b, c, d: self.sample_latent(a, b, c, d), device=dev)
            self.decoder.share_side_stream(None)
        return i2v_dist.OverlappedCollator(total, group=group, stream=self._prefetch.stream)

    def check(self):
        """Raises if the decoder's split-fp16 operands left the fp16 range in any call since the last check (sticky device
        flag, i2v_dec_status).  Synchronises: call it where the results are brought to the host anyway."""
        fb = self.decoder.native().fallback_layers()
        if fb["layers"] or fb["whole_handle"]:
            import warnings
            warnings.warn("decoder (mma = auto): the range guard switched " + ("the whole handle" if fb["whole_handle"] else ", ".join(fb["layers"])) +
                          f" to the exact-fp32 kernels ({fb['reruns']} forward(s) were run again): this checkpoint's activations leave the window "
                          "the split-fp16 operand format holds 1e-4 in (INTEGRATION.md §3); the frames are valid", RuntimeWarning)
        flags = self.decoder.native().status()
        if flags & 2 and not flags & 1:
            import warnings
            self.decoder.native().status(reset=True)
            warnings.warn("decoder status bit 1 (underflow): a conv operand tensor lay entirely below 2^-10, where the split-fp16 "
                          "format no longer holds 1e-4 relative L2 (INTEGRATION.md §3); the frames are finite but less precise -- "
                          "use Generator(dic['mma'] = 0) / I2V_DEC_MMA=0 for this checkpoint", RuntimeWarning)
            return
        if flags:
            raise RuntimeError(f"decoder status flags {flags}: activations left the fp16 range of the split-fp16 conv operands -- "
                               "the frames of this call are invalid; use Generator(dic['mma'] = 0) / I2V_DEC_MMA=0 for this checkpoint")

    def forward(self, x_0, cond=None, residual=None, embed=None):
        """Input: x_0 (start frame) of shape (BS, C, H, W).  Output as the reference: ``seq[:vid_length]`` --
        a slice over the BATCH dimension (quirk Q3)."""
        return self.synthesize(x_0, cond, residual, embed)[:self.vid_length]

    @torch.no_grad()
    def transfer(self, seq_query, x_0, embed_query=None, embed=None):
        """Motion transfer (reference get_model.py:77-103).  seq_query [BS,T,C,H,W]; x_0 [BS',C,H,W] start frames the motion is
        transferred to.  Returns the un-sliced sequence [BS',T',C,H,W].  ``embed_query`` / ``embed`` optionally replace the
        conditioning embedder's outputs for seq_query[:, 0] and x_0."""
        _, z, _ = self.encoder(seq_query[:, 1:].transpose(1, 2))                       # :87 (the MEAN is kept)
        res, _ = self.flow(z, [seq_query[:, 0].contiguous()], embed=embed_query)       # :90 cINN forward
        res = res.view(z.size(0), -1).repeat(x_0.size(0), 1).contiguous()
        z_ref = self.flow(res, [x_0], reverse=True, embed=embed).view(x_0.size(0), -1)  # :93
        return self.decoder.decode_sequence(x_0, z_ref, self.vid_length)               # :96-101
