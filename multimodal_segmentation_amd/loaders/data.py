"""Single-modality slice container with the interface of reference loaders/data.py:13-171: images [N,H,W,1], masks
[N,H,W,L] and, per slice, the id of the volume it came from.  Host-side numpy.  DICOM / NIfTI reading (loaders/chaos.py)
is out of scope: containers are filled from arrays (synthetic volumes).

Every selection operator is expressed through one primitive, `_take(rows)`, which re-indexes the three arrays together;
random draws use the global numpy RNG in the same order as the reference so that a fixed seed selects the same slices."""
import logging
import os

import numpy as np

from ..utils import data_utils

log = logging.getLogger('data')


def block_mean(a, ratio):
    """Mean over ratio x ratio spatial blocks, zero-padded to a multiple of the ratio -- what
    skimage.measure.block_reduce(a, (1, r, r, 1), np.mean) computes (data.py:155-160)."""
    n, h, w, c = a.shape
    ph, pw = (-h) % ratio, (-w) % ratio
    if ph or pw:
        a = np.pad(a, ((0, 0), (0, ph), (0, pw), (0, 0)), 'constant')
    return a.reshape(n, (h + ph) // ratio, ratio, (w + pw) // ratio, ratio, c).mean(axis=(2, 4))


def _seed(seed):
    if seed > -1:
        np.random.seed(seed)


class Data(object):
    def __init__(self, images, masks, index, downsample=1):
        assert images.shape[:-1] == masks.shape[:-1] and images.shape[0] == index.shape[0]
        self.image_shape, self.mask_shape = images.shape[1:], masks.shape[1:]
        self.images, self.masks, self.index = images, masks, index
        self.num_volumes = len(self.volumes())
        self.downsample(downsample)
        log.info('Data: images %s, %d volumes, range [%.1f, %.1f]' % (str(self.images.shape), self.num_volumes,
                                                                      images.min() if images.size else 0,
                                                                      images.max() if images.size else 0))

    # ---- primitives ----------------------------------------------------------------------------------------------
    def _take(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        self.images, self.masks, self.index = self.images[rows], self.masks[rows], self.index[rows]

    def _rows_of(self, vol):
        return np.nonzero(self.index == vol)[0]

    def volumes(self):
        return sorted(set(np.asarray(self.index).tolist()))

    def size(self):
        return len(self.images)

    def shape(self):
        return self.image_shape

    def get_images(self, vol):
        return self.images[self.index == vol]

    def get_masks(self, vol):
        return self.masks[self.index == vol]

    def copy(self):
        return Data(np.copy(self.images), np.copy(self.masks), np.copy(self.index))

    # ---- combination / geometry ------------------------------------------------------------------------------------
    def merge(self, other):
        for mine, theirs in ((self.images, other.images), (self.masks, other.masks)):
            assert mine.shape[1:] == theirs.shape[1:], '%s vs %s' % (mine.shape, theirs.shape)
        self.images = np.concatenate([self.images, other.images], axis=0)
        self.masks = np.concatenate([self.masks, other.masks], axis=0)
        self.index = np.concatenate([self.index, other.index], axis=0)
        self.num_volumes = len(self.volumes())

    def crop(self, shape):
        [self.images], [self.masks] = data_utils.crop_same([self.images], [self.masks], size=shape, pad_mode='constant')
        assert self.images.shape[1:-1] == self.masks.shape[1:-1] == tuple(shape)

    def downsample(self, ratio=2):
        if ratio == 1:
            return
        self.images = block_mean(self.images, ratio)
        if self.masks is not None:
            self.masks = block_mean(self.masks, ratio)

    # ---- selection (global numpy RNG, reference draw order) ---------------------------------------------------------
    def shuffle(self):
        rows = np.arange(self.size())
        np.random.shuffle(rows)
        self._take(rows)

    def sample_per_volume(self, num, seed=-1):
        """`num` slices from every volume without replacement (data.py:84-113).  A volume shorter than `num` is kept
        whole but -- as in the reference -- still gets `num` index entries, which leaves the index longer than the
        arrays; callers there never hit that case and neither should callers here."""
        _seed(seed)
        rows, index = [], []
        for vol in self.volumes():
            own = self._rows_of(vol)
            pick = np.arange(len(own)) if len(own) < num else np.random.choice(len(own), size=num, replace=False)
            rows.append(own[pick])
            index.append(np.array([vol] * num))
        rows = np.concatenate(rows)
        self.images, self.masks = self.images[rows], self.masks[rows]
        self.index = np.concatenate(index, axis=0)

    def sample_images(self, num, seed=-1):
        _seed(seed)
        self._take(np.random.choice(self.size(), size=num, replace=False))

    def get_sample_volumes(self, num, seed=-1):
        _seed(seed)
        return np.random.choice(self.volumes(), size=num, replace=False)

    def sample(self, num, seed=-1):
        """keep `num` randomly chosen volumes (no draw at all when every volume is kept, data.py:131-136)"""
        if num != self.num_volumes:
            self.filter_volumes(self.get_sample_volumes(num, seed))

    def filter_volumes(self, volumes):
        """keep the listed volumes, in the listed order (data.py:138-150)"""
        rows = [self._rows_of(v) for v in volumes]
        self._take(np.concatenate(rows) if rows else np.zeros((0,), np.int64))
        self.num_volumes = len(volumes)

    def save(self, folder):
        os.makedirs(folder, exist_ok=True)
        for i in range(self.size()):
            np.savez_compressed(folder + '/images_%d' % i, self.images[i:i + 1])
            np.savez_compressed(folder + '/masks_%d' % i, self.masks[i:i + 1])
