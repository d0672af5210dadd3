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
from stage2_cINN.modules import INN


class Model(torch.nn.Module):
    def __init__(self, model_path, vid_length, transfer=False, embedder=None):
        super().__init__()
        if transfer:
            raise NotImplementedError("Model(transfer=True): the 3D-ResNet motion encoder is row N3 of the coverage "
                                      "contract (SURVEY §8f), not part of the sampling hot path")
        opt = i2v_config.load(os.path.join(model_path, "config_stage2.yaml"))
        path_stage1 = opt.First_stage_model["model_path"] + opt.First_stage_model["model_name"] + "/"
        config = i2v_config.load(path_stage1 + "config_stage1.yaml")

        self.decoder = decoder.Generator(config.Decoder).cuda()
        self.decoder.load_state_dict(torch.load(path_stage1 + opt.First_stage_model["checkpoint_decoder"] + ".pth",
                                                map_location="cpu")["state_dict"])
        _ = self.decoder.eval()

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

    @torch.no_grad()
    def synthesize(self, x_0, cond=None, residual=None, embed=None):
        """[B,3,H,W] -> [B, 16*ceil(vid_length/16), 3, H, W]; no batch slice."""
        if residual is None:
            residual = torch.randn(x_0.size(0), self.z_dim).cuda()  # CPU generator, like get_model.py:59
        cond = [x_0, cond]
        z = self.flow(residual, cond, reverse=True, embed=embed).view(x_0.size(0), -1)
        seq = self.decoder(x_0, z)
        while seq.shape[1] < self.vid_length:
            seq1 = self.decoder(seq[:, -1].contiguous(), z)
            seq = torch.cat((seq, seq1), dim=1)
        return seq

    def forward(self, x_0, cond=None, residual=None, embed=None):
        """Input: x_0 (start frame) of shape (BS, C, H, W).  Output as the reference: ``seq[:vid_length]`` --
        a slice over the BATCH dimension (quirk Q3)."""
        return self.synthesize(x_0, cond, residual, embed)[:self.vid_length]
