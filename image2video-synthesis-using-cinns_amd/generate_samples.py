#!/usr/bin/env python
"""Sampling CLI, mirror of the reference's ``generate_samples.py``:

    python generate_samples.py -gpu 0 -dataset bair [-texture fire] [-ckpt_path DIR/] [-seq_length 16] [-bs 6]

globs ``./assets/GT_samples/<dataset>/*.{jpg,png,jpeg}``, normalises to [-1,1], resizes to ``Data.img_size``, runs
``Model`` batch by batch and writes ``./assets/results/<dataset>/results.gif``.

The image I/O side is thin glue: cv2 / kornia / imageio are replaced by PIL + numpy + a bilinear
``F.interpolate`` (align_corners=False, kornia.Resize's default).  The conditioning embedding comes from the ResNet-50
embedder that ``Model`` loads from ``Conditioning_Model.model_path`` (row N1, csrc/i2v_embed.hip), exactly as in the
reference; for checkpoints shipped without it pass ``-embed_npy FILE`` with a precomputed ``[N,E]`` embedding, or
``-embed_seed S`` to draw a synthetic one (demo / smoke use).  E is the embedder's width (``Conditioning_Model.z_dim``):
for endpoint-controlled models the 30 position one-hots are appended by the model, not by the caller.
"""
import argparse
import glob
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

img_suffix = ["jpg", "png", "jpeg"]


def load_images(names, img_res):
    from PIL import Image
    imgs = []
    for name in names:
        a = np.asarray(Image.open(name).convert("RGB"), dtype=np.float32) / 255.0      # HWC RGB in [0,1]
        t = torch.from_numpy(a).permute(2, 0, 1)[None]
        t = (t - 0.5) / 0.5                                                           # Normalize(0.5, 0.5)
        imgs.append(F.interpolate(t, size=(img_res, img_res), mode="bilinear", align_corners=False))
    return torch.cat(imgs)


def save_gif(path, frames, fps=3):
    from PIL import Image
    ims = [Image.fromarray(f) for f in frames.astype(np.uint8)]
    ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(1000 / fps), loop=0)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-gpu", type=str, required=True, help="Define GPU on which to run")
    parser.add_argument("-dataset", type=str, required=True, help="Specify dataset")
    parser.add_argument("-texture", type=str, help="Specify texture when using DTDB")
    parser.add_argument("-ckpt_path", type=str, required=False, help="If ckpt outside of repo")
    parser.add_argument("-seq_length", type=int, default=16)
    parser.add_argument("-bs", type=int, default=6, help="Batchsize")
    parser.add_argument("-embed_npy", type=str, help="[N,E] conditioning embeddings (one row per image)")
    parser.add_argument("-embed_seed", type=int, help="draw synthetic conditioning embeddings with this seed")
    parser.add_argument("-img_path", type=str, help="override ./assets/GT_samples/<dataset>/")
    parser.add_argument("-out_path", type=str, help="override ./assets/results/<dataset>/")
    parser.add_argument("-seed", type=int, help="seed the CPU generator the latent residuals are drawn from, right before sampling")
    parser.add_argument("-raw_npy", type=str, help="also write the uint8 frame strip [T,H,N*W,3] (the GIF's palette is lossy)")
    args = parser.parse_args(argv)
    os.environ["HIP_VISIBLE_DEVICES"] = args.gpu   # the reference sets CUDA_VISIBLE_DEVICES (generate_samples.py:20)

    from get_model import Model
    from utils import auxiliaries as aux

    path_ds = f"{args.dataset}/{args.texture}" if args.dataset == "DTDB" else f"{args.dataset}"
    ckpt_path = f"./models/{path_ds}/stage2/" if not args.ckpt_path else args.ckpt_path
    img_path = args.img_path or f"./assets/GT_samples/{path_ds}/"
    img_list = []
    for suffix in img_suffix:
        img_list.extend(sorted(glob.glob(img_path + f"*.{suffix}")))
    if not img_list:
        raise SystemExit(f"no images found under {img_path}")

    model = Model(ckpt_path, args.seq_length)
    img_res = model.config.Data["img_size"]
    imgs = load_images(img_list, img_res)
    E = model.flow.flow.cond_channels - 3 * model.flow.cond_size  # width of the image embedding (without the control one-hots)
    if args.embed_npy:
        embeds = torch.from_numpy(np.load(args.embed_npy).astype(np.float32))
    elif args.embed_seed is not None:
        embeds = torch.randn(imgs.size(0), E, generator=torch.Generator().manual_seed(args.embed_seed))
    else:
        embeds = None  # Model raises with a clear message unless an embedder object was attached

    if args.seed is not None:
        torch.manual_seed(args.seed)   # (Model.forward draws torch.randn on the global CPU generator, get_model.py:59)
    bs = args.bs
    length = math.ceil(imgs.size(0) / bs)
    videos = []
    from i2v_pipeline import LatentPrefetcher
    with torch.no_grad():
        # the cINN pass of batch i + 1 runs on a side stream underneath the decoder of batch i; the residual draws keep
        # the order of the reference's loop (generate_samples.py:44-54)
        def inputs(i):
            batch = imgs[i * bs:(i + 1) * bs].cuda()
            return batch, (embeds[i * bs:(i + 1) * bs].cuda() if embeds is not None else None)
        pf = LatentPrefetcher(lambda b, e: model.sample_latent(b, embed=e))
        batch, emb = inputs(0)
        ticket = pf.submit(batch, emb)
        for i in range(length):
            z = pf.get(ticket)
            cur = batch
            if i + 1 < length:
                batch, emb = inputs(i + 1)
                ticket = pf.submit(batch, emb)
            videos.append(model.decode(cur, z)[:model.vid_length].cpu())   # [:vid_length]: the BATCH slice of get_model.py:75 (Q3)
            model.check()   # the .cpu() above synchronised: a range overflow of this batch is reported now, not a call later
    videos = torch.cat(videos)

    save_path = args.out_path or f"./assets/results/{path_ds}/"
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    gif = aux.convert_seq2gif(videos)
    save_gif(save_path + "results.gif", gif, fps=3)
    if args.raw_npy:
        np.save(args.raw_npy, gif.astype(np.uint8))
    print(f"Animations saved in {save_path}")


if __name__ == "__main__":
    main()
