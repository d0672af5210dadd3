#!/usr/bin/env python
"""Motion-transfer CLI, mirror of the reference's ``generate_transfer.py``:

    python generate_transfer.py -gpu 0 -dataset landscape [-ckpt_path DIR/] [-seq_length 16] [-bs 6]

Every sub-folder of ``./assets/GT_samples/<dataset>/transfer/`` is a query clip (its first ``seq_length`` frames); the
motion of each query is transferred to the first frame of every clip (``Model.transfer``, get_model.py:77-103) and written
to ``./assets/results/<dataset>/transfer_<idx>.gif`` with the query clip in the first column.
Image I/O as in generate_samples.py (PIL + numpy instead of cv2 / kornia / imageio / natsort)."""
import argparse
import glob
import math
import os

import torch

from generate_samples import img_suffix, load_images, save_gif


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-gpu", type=str, required=True, help="Define GPU on which to run")
    parser.add_argument("-dataset", type=str, required=True, help="Specify dataset")
    parser.add_argument("-ckpt_path", type=str, required=False)
    parser.add_argument("-seq_length", type=int, default=16)
    parser.add_argument("-bs", type=int, default=6, help="Batchsize")
    parser.add_argument("-img_path", type=str, help="override ./assets/GT_samples/<dataset>/transfer/")
    parser.add_argument("-out_path", type=str, help="override ./assets/results/<dataset>/")
    args = parser.parse_args(argv)
    os.environ["HIP_VISIBLE_DEVICES"] = args.gpu

    from get_model import Model
    from utils import auxiliaries as aux

    ckpt_path = f"./models/{args.dataset}/stage2/" if not args.ckpt_path else args.ckpt_path
    model = Model(ckpt_path, args.seq_length, transfer=True)
    img_path = args.img_path or f"./assets/GT_samples/{args.dataset}/transfer/"
    img_res = model.config.Data["img_size"]
    videos = []
    for vidp in sorted(os.listdir(img_path)):
        img_list = []
        for suffix in img_suffix:
            img_list.extend(glob.glob(img_path + vidp + "/" + f"*.{suffix}"))
        img_list = sorted(img_list)[:args.seq_length]
        if img_list:
            videos.append(load_images(img_list, img_res))
    if not videos:
        raise SystemExit(f"no clips found under {img_path}")
    videos = torch.stack(videos)                                   # [N, T, 3, H, W]

    bs = args.bs
    length = math.ceil(videos.size(0) / bs)
    save_path = args.out_path or f"./assets/results/{args.dataset}/"
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    for idx, query in enumerate(videos):
        transfer = []
        with torch.no_grad():
            for i in range(length):
                batch = videos[i * bs:(i + 1) * bs, 0].cuda()
                transfer.append(model.transfer(query[None, :].cuda(), batch).cpu())
                model.check()   # (synchronised by .cpu(): a range overflow of the split-fp16 operands is reported for THIS call)
        transfer = torch.cat(transfer)
        t = min(transfer.shape[1], query.shape[0])
        transfer = torch.cat((query[None, :t], transfer[:, :t]), dim=0)
        save_gif(save_path + f"transfer_{idx}.gif", aux.convert_seq2gif(transfer), fps=3)
    print(f"Animations saved in {save_path}")


if __name__ == "__main__":
    main()
