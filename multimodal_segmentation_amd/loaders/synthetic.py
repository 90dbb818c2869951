"""Seeded synthetic stand-in for the CHAOS T1/T2 volumes: there is no dataset in the container or on
the GPU box, and the benchmark metric is defined on synthetic slices.  Mirrors the container API the executors use
of reference loaders/MultimodalPairedData.py (get_images_modi / get_masks_modi / num_volumes / size) on
14/3/3 "volumes" x 20 slices (split sizes of reference loaders/chaos.py:34-37)."""
import numpy as np


def smooth_field(rng, H, W, sigma):
    from scipy.ndimage import gaussian_filter
    f = np.tanh(gaussian_filter(rng.standard_normal((H, W)), sigma) * 6.0)
    f = (f - f.min()) / (f.max() - f.min() + 1e-12)
    return (f * 2 - 1).astype(np.float32)       # rescaled per slice to exactly [-1, 1] (chaos.py:242-246)


def ellipse_masks(rng, H, W, num_masks):
    yy, xx = np.mgrid[:H, :W]
    out = np.zeros((H, W, num_masks), np.float32)
    taken = np.zeros((H, W), bool)
    for k in range(num_masks):
        cy, cx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W
        ry, rx = rng.uniform(0.06, 0.18) * H, rng.uniform(0.06, 0.18) * W
        m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0) & ~taken
        taken |= m
        out[..., k] = m
    return out


class SyntheticPairedData(object):
    """Two modalities of the same synthetic anatomy: modality 2 is a smooth intensity remap of modality 1 plus its
    own texture, organs are brighter/darker ellipses, so that segmentation is learnable."""

    def __init__(self, input_shape, num_masks, volumes, slices_per_volume, seed):
        H, W = input_shape[0], input_shape[1]
        rng = np.random.RandomState(seed)
        n = len(volumes) * slices_per_volume
        self.images = [np.zeros((n, H, W, 1), np.float32) for _ in range(2)]
        self.masks = [np.zeros((n, H, W, num_masks), np.float32) for _ in range(2)]
        self.index = np.repeat(np.asarray(volumes), slices_per_volume)
        sigma = max(H / 32.0, 1.0)
        for i in range(n):
            m = ellipse_masks(rng, H, W, num_masks)
            organ = (m * np.linspace(0.4, 1.0, num_masks)[None, None]).sum(-1)
            for mod in range(2):
                tex = smooth_field(rng, H, W, sigma)
                img = 0.5 * tex + (organ if mod == 0 else -organ)
                img = (img - img.min()) / (img.max() - img.min() + 1e-12) * 2 - 1
                self.images[mod][i, ..., 0] = img
                self.masks[mod][i] = m

    def get_images_modi(self, mod_i):
        return self.images[mod_i]

    def get_masks_modi(self, mod_i):
        return self.masks[mod_i]

    def volumes(self):
        return sorted(set(self.index.tolist()))

    @property
    def num_volumes(self):
        return len(self.volumes())

    def size(self):
        return self.images[0].shape[0]

    def get_volume(self, mod_i, vol):
        sel = self.index == vol
        return self.images[mod_i][sel], self.masks[mod_i][sel]


def splits():
    """14 train / 3 validation / 3 test volumes"""
    return {'training': list(range(14)), 'validation': list(range(14, 17)), 'test': list(range(17, 20))}
