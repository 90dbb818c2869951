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

    # ---- container API of reference loaders/MultimodalPairedData.py used by the executors / tester ---------------
    def get_volume_images_modi(self, mod_i, vol):
        return self.images[mod_i][self.index == vol]

    def get_volume_masks_modi(self, mod_i, vol):
        return self.masks[mod_i][self.index == vol]

    def set_images_modi(self, mod_i, images):
        self.images[mod_i] = images

    def set_masks_modi(self, mod_i, masks):
        self.masks[mod_i] = masks

    def filter_volumes(self, volumes):
        """keep only the given volumes (MultimodalPairedData.py:46-62)"""
        sel = np.isin(self.index, np.asarray(volumes))
        self.images = [a[sel] for a in self.images]
        self.masks = [a[sel] for a in self.masks]
        self.index = self.index[sel]

    def crop(self, shape):
        """centre crop to `shape` (MultimodalPairedData.py:64-72); synthetic data is generated at the target size"""
        H, W = self.images[0].shape[1:3]
        if (H, W) == tuple(shape):
            return
        t, l = (H - shape[0]) // 2, (W - shape[1]) // 2
        self.images = [a[:, t:t + shape[0], l:l + shape[1]] for a in self.images]
        self.masks = [a[:, t:t + shape[0], l:l + shape[1]] for a in self.masks]

    def sample_images(self, num, seed=-1):
        if seed > -1:
            np.random.seed(seed)
        idx = np.random.choice(self.size(), size=num, replace=False)
        return [a[idx] for a in self.images]

    def randomise_pairs(self, length=3, seed=None):
        """Re-pair every modality-2 slice with a modality-1 slice at most `length` slices away inside its volume
        (MultimodalPairedData.py:143-167)."""
        if seed is not None:
            np.random.seed(seed)
        new_images, new_masks = self.images[0].copy(), self.masks[0].copy()
        for vol in self.volumes():
            pos = np.where(self.index == vol)[0]
            n = len(pos)
            offsets = np.random.randint(-length, length, size=n) if length > 0 else np.zeros(n, int)
            tgt = np.clip(np.arange(n) + offsets, 0, n - 1)
            new_images[pos] = self.images[0][pos[tgt]]
            new_masks[pos] = self.masks[0][pos[tgt]]
        self.images[0], self.masks[0] = new_images, new_masks

    def copy(self):
        import copy
        c = copy.copy(self)
        c.images = [a.copy() for a in self.images]
        c.masks = [a.copy() for a in self.masks]
        c.index = self.index.copy()
        return c

    def merge(self, other):
        self.images = [np.concatenate([a, b], 0) for a, b in zip(self.images, other.images)]
        self.masks = [np.concatenate([a, b], 0) for a, b in zip(self.masks, other.masks)]
        self.index = np.concatenate([self.index, other.index], 0)


def splits():
    """14 train / 3 validation / 3 test volumes"""
    return {'training': list(range(14)), 'validation': list(range(14, 17)), 'test': list(range(17, 20))}
