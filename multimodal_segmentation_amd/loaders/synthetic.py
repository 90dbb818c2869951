"""Seeded synthetic stand-in for the CHAOS T1/T2 volumes: there is no dataset in the container or on
the GPU box, and the benchmark metric is defined on synthetic slices.  Mirrors the container API the executors use
of reference loaders/MultimodalPairedData.py (get_images_modi / get_masks_modi / num_volumes / size) on
14/3/3 "volumes" x 20 slices (split sizes of reference loaders/chaos.py:34-37)."""
import numpy as np

from .MultimodalPairedData import MultimodalPairedData


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


class SyntheticPairedData(MultimodalPairedData):
    """Two (or `num_modalities`) modalities of the same synthetic anatomy: modality 2 is a smooth intensity remap of modality 1
    plus its own texture, organs are brighter/darker ellipses, so that segmentation is learnable; a third modality (BASELINE
    config #5, a build-defined extension) shows the organs with alternating sign."""

    def __init__(self, input_shape, num_masks, volumes, slices_per_volume, seed, num_modalities=2):
        H, W = input_shape[0], input_shape[1]
        rng = np.random.RandomState(seed)
        n = len(volumes) * slices_per_volume
        M = int(num_modalities)
        images = np.zeros((n, H, W, M), np.float32)
        masks = np.zeros((n, H, W, M * num_masks), np.float32)
        index = np.repeat(np.asarray(volumes), slices_per_volume)
        sigma = max(H / 32.0, 1.0)
        for i in range(n):
            m = ellipse_masks(rng, H, W, num_masks)
            organ = (m * np.linspace(0.4, 1.0, num_masks)[None, None]).sum(-1)
            alt = (m * (np.linspace(0.4, 1.0, num_masks) * np.where(np.arange(num_masks) % 2 == 0, 1.0, -1.0))[None, None]).sum(-1)
            for mod in range(M):
                tex = smooth_field(rng, H, W, sigma)
                img = 0.5 * tex + (organ if mod == 0 else (-organ if mod == 1 else alt))
                images[i, ..., mod] = (img - img.min()) / (img.max() - img.min() + 1e-12) * 2 - 1
                masks[i, ..., mod * num_masks:(mod + 1) * num_masks] = m
        super(SyntheticPairedData, self).__init__(images, masks, index, num_modalities=M)
        self.image_dict = {k: np.ascontiguousarray(v) for k, v in self.image_dict.items()}
        self.masks_dict = {k: np.ascontiguousarray(v) for k, v in self.masks_dict.items()}

    def get_volume(self, mod_i, vol):
        return self.get_volume_images_modi(mod_i, vol), self.get_volume_masks_modi(mod_i, vol)


def splits():
    """14 train / 3 validation / 3 test volumes"""
    return {'training': list(range(14)), 'validation': list(range(14, 17)), 'test': list(range(17, 20))}
