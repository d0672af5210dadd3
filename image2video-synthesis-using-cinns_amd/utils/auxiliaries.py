"""The two helpers of the reference's ``utils/auxiliaries.py`` that the sampling CLI uses (:15-22, :53-55)."""
import numpy as np


def denorm(x):
    out = (x + 1) / 2
    return out.clamp_(0, 1)


def convert_seq2gif(sequence):
    """[N,T,3,H,W] in [-1,1] -> [T,H,N*W,3] float array scaled to 0..255 by its own maximum (reference :15-22)."""
    img_shape = sequence.shape
    images_orig = denorm(sequence).permute(0, 1, 3, 4, 2).detach().cpu().numpy()
    img_gif = images_orig[0]
    for i in range(1, img_shape[0]):
        img_gif = np.concatenate((img_gif, images_orig[i]), axis=2)
    img_gif = 255 * img_gif / np.max(img_gif)
    return img_gif
