"""Mirror of the reference's ``stage2_cINN/modules/INN.py``: ``SupervisedTransformer`` owns the conditional flow
and the frozen conditioning embedder and dispatches forward / reverse (reference INN.py:8-73).

The ResNet-50 conditioning embedder (stage2_cINN/AE/modules/AE.py:91-166, row N1) is built like the reference does
(INN.py:36-41) from ``dic['model_path'] + dic['model_name'] + '/config_stage2_AE.yaml'`` and ``<checkpoint_name>.pth``
when those files exist; otherwise the embedding has to be supplied by the caller (``embed=``) or through an injected
``embedder`` object exposing ``encode(x).mode()``."""
import os

import numpy as np
import torch
import torch.nn as nn

import i2v_config
from stage2_cINN.AE.modules.AE import ResnetEncoder
from stage2_cINN.modules.flow_blocks import ConditionalFlow


class SupervisedTransformer(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        in_channels = kwargs["flow_in_channels"]
        mid_channels = kwargs["flow_mid_channels"]
        hidden_depth = kwargs["flow_hidden_depth"]
        n_flows = kwargs["n_flows"]
        conditioning_option = kwargs["flow_conditioning_option"]
        embedding_channels = kwargs["flow_embedding_channels"] if "flow_embedding_channels" in kwargs \
            else kwargs["flow_in_channels"]
        self.control = bool(kwargs["control"])
        self.cond_size = 10 if self.control else 0
        self.flow = ConditionalFlow(in_channels=in_channels, embedding_dim=embedding_channels + self.cond_size * 3,
                                    hidden_dim=mid_channels, hidden_depth=hidden_depth, n_flows=n_flows,
                                    conditioning_option=conditioning_option, control=self.control)
        self.embedder = kwargs.get("embedder", None)
        dic = kwargs.get("dic", None)
        if self.embedder is None and dic is not None and dic.get("model_path") is not None:
            model_path = dic["model_path"] + dic["model_name"] + "/"                      # INN.py:37
            cfg, ckpt = model_path + "config_stage2_AE.yaml", model_path + dic["checkpoint_name"] + ".pth"
            if os.path.exists(cfg) and os.path.exists(ckpt):
                config = i2v_config.load(cfg)                                              # INN.py:38
                self.embedder = ResnetEncoder(config.AE)                                   # INN.py:39
                self.embedder.load_state_dict(torch.load(ckpt, map_location="cpu")["state_dict"])  # INN.py:40
                _ = self.embedder.eval()

    def embed_pos(self, pos):
        """Three one-hots of 10 bins at index floor(pos*10 - 1e-4) (reference INN.py:49-57)."""
        pos = pos.detach().float().cpu() * self.cond_size - 1e-4
        out = torch.zeros(pos.size(0), 3 * self.cond_size)
        rows = np.arange(pos.size(0))
        for j in range(3):
            out[rows, j * self.cond_size + pos[:, j].long()] = 1
        return out

    def _embed(self, input, cond, embed):
        if embed is None:
            if self.embedder is None:
                raise RuntimeError("SupervisedTransformer: no conditioning embedder is attached (its config / checkpoint "
                                   "were not found under Conditioning_Model.model_path); pass embed=[B,E] explicitly")
            with torch.no_grad():
                embed = self.embedder.encode(cond[0]).mode().reshape(input.size(0), -1).detach()
        if self.control and embed.shape[1] == self.flow.cond_channels - 3 * self.cond_size:
            # (a caller-supplied embedding may already carry the 30 position one-hots: full width -> used as is)
            embed = torch.cat((embed, self.embed_pos(cond[1]).to(embed)), dim=1)
        return embed.contiguous()

    def forward(self, input, cond, reverse=False, train=False, embed=None):
        embed = self._embed(input, cond, embed)
        if reverse:
            return self.reverse(input, embed)
        out, logdet = self.flow(input, embed)
        return out, logdet

    def reverse(self, out, cond):
        return self.flow(out, cond, reverse=True)

    def sample(self, shape, cond):
        """reference INN.py:43-47 (calls reverse with the raw cond, i.e. cond must already be an embedding)."""
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        z_tilde = torch.randn(shape).to(device)
        return self.reverse(z_tilde, cond)
