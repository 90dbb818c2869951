"""Paired two-modality slice container with the interface of reference loaders/MultimodalPairedData.py:8-188.

Constructed like the reference from images / masks concatenated on the channel axis ([N,H,W,2] and [N,H,W,2L]); kept as
one (images, masks) pair of arrays per modality that share a single volume index.  The pairing operators used by the
randomised-pairs and automated-pairing experiments (`randomise_pairs`, `expand_pairs`) are split into a pure index
computation (`random_pair_index`, `neighbour_window`) and its application, so the index logic is testable by itself;
random draws use the global numpy RNG in the reference's order."""
import logging

import numpy as np

from ..utils import data_utils
from .data import Data

log = logging.getLogger('MultimodalPairedData')


class MultimodalPairedData(Data):
    def __init__(self, images, masks, index, downsample=1, num_modalities=2):
        """`num_modalities` > 2 is a build-defined extension (BASELINE config #5); the reference splits into two halves
        (MultimodalPairedData.py:14-22)."""
        super(MultimodalPairedData, self).__init__(images, masks, index, downsample)
        stacked_images, stacked_masks = self.images, self.masks
        del self.images, self.masks                      # per-modality storage from here on, as in the reference
        M = int(num_modalities)
        assert stacked_images.shape[-1] == M and stacked_masks.shape[-1] % M == 0, (stacked_images.shape, stacked_masks.shape, M)
        self.num_modalities = M
        self.masks_per_mod = k = stacked_masks.shape[-1] // M
        self.image_dict = {m: stacked_images[..., m:m + 1] for m in range(M)}
        self.masks_dict = {m: stacked_masks[..., m * k:(m + 1) * k] for m in range(M)}

    # ---- accessors ---------------------------------------------------------------------------------------------------
    def get_images_modi(self, mod_i):
        return self.image_dict[mod_i]

    def get_masks_modi(self, mod_i):
        return self.masks_dict[mod_i]

    def set_images_modi(self, mod_i, images):
        self.image_dict[mod_i] = images

    def set_masks_modi(self, mod_i, masks):
        self.masks_dict[mod_i] = masks

    def get_volume_images_modi(self, mod_i, vol):
        return self.image_dict[mod_i][self.index == vol]

    def get_volume_masks_modi(self, mod_i, vol):
        return self.masks_dict[mod_i][self.index == vol]

    def size(self):
        return int(max(self.image_dict[m].shape[0] for m in range(self.num_modalities)))

    def _take(self, rows):
        """re-index every modality and the volume index together"""
        rows = np.asarray(rows, dtype=np.int64)
        for m in range(self.num_modalities):
            self.image_dict[m] = self.image_dict[m][rows]
            self.masks_dict[m] = self.masks_dict[m][rows]
        self.index = self.index[rows]

    def copy(self):
        c = object.__new__(type(self))
        c.__dict__.update(self.__dict__)
        c.image_dict = {m: np.copy(a) for m, a in self.image_dict.items()}      # keeps expanded neighbourhoods, if any
        c.masks_dict = {m: np.copy(a) for m, a in self.masks_dict.items()}
        c.index = np.copy(self.index)
        return c

    def merge(self, other):
        for m in range(self.num_modalities):
            self.image_dict[m] = np.concatenate([self.image_dict[m], other.get_images_modi(m)], axis=0)
            self.masks_dict[m] = np.concatenate([self.masks_dict[m], other.get_masks_modi(m)], axis=0)
        self.index = np.concatenate([self.index, other.index], axis=0)
        assert self.image_dict[0].shape[0] == self.index.shape[0]
        self.num_volumes = len(self.volumes())

    def crop(self, shape):
        for m in range(self.num_modalities):
            [img], [msk] = data_utils.crop_same([self.image_dict[m]], [self.masks_dict[m]], size=shape, pad_mode='constant')
            assert img.shape[1:-1] == msk.shape[1:-1] == tuple(shape), (img.shape, msk.shape, shape)
            self.image_dict[m], self.masks_dict[m] = img, msk

    def sample_images(self, num, seed=-1):
        """MultimodalPairedData.py:77-90.  (The reference calls its setters without the modality index there and cannot
        run; the evident intent -- one draw applied to every modality -- is what Data.sample_images + _take do.)"""
        super(MultimodalPairedData, self).sample_images(num, seed)

    # filter_volumes / sample / get_sample_volumes / shuffle are inherited: they go through _take.

    # ---- pairing operators ---------------------------------------------------------------------------------------------
    @staticmethod
    def neighbour_window(i, n_other, n_own, offsets):
        """Candidate partners of slice i (MultimodalPairedData.py:107-118): 2*offsets+1 consecutive slices clamped inside
        the volume (a shorter volume is padded with slice 0), re-ordered so that the expert partner i comes first."""
        width = 2 * offsets + 1
        if n_own < width:
            window = list(range(n_own)) + [0] * (width - n_own)
        else:
            start = min(max(i - offsets, 0), n_other - width)
            window = list(range(start, start + width))
        window.remove(i)
        return [i] + window

    def expand_pairs(self, offsets, mod_i, neighborhood=2):
        """Replace modality `mod_i`'s images [N,H,W,1] by [N,H,W,neighborhood]: channel 0 is the expert partner, the rest
        are drawn without replacement from the +-offsets window (MultimodalPairedData.py:92-141).  In place."""
        assert mod_i in [0, 1], 'mod_i selects the modality whose neighbourhood is enlarged'
        columns = []
        for vol in self.volumes():
            rows = np.nonzero(self.index == vol)[0]
            n = len(rows)                                # both modalities hold the same slices of a volume
            for i in range(n):
                window = self.neighbour_window(i, n, n, offsets)
                if len(window) > neighborhood:
                    window = [window[0]] + list(np.random.choice(window[1:], size=neighborhood - 1, replace=False))
                assert len(window) <= neighborhood, 'Exceeded maximum neighborhood size'
                columns.append(rows[np.asarray(window, dtype=np.int64)])
        columns = np.stack(columns, axis=0)              # [N, neighborhood] source rows
        assert columns.shape[-1] == neighborhood, '%s vs %s' % (columns.shape[-1], neighborhood)
        src = self.image_dict[mod_i][..., 0]
        self.image_dict[mod_i] = np.stack([src[columns[:, j]] for j in range(neighborhood)], axis=-1)

    @staticmethod
    def random_pair_index(n, length):
        """Partner row of every slice of an n-slice volume (MultimodalPairedData.py:152-161): offsets in [-length, length),
        re-drawn for the first `length` slices when they point before the volume and for the last `length-1` slices when
        they point past its end."""
        offsets = np.random.randint(-length, length, size=n)
        for head in range(length):
            if head + offsets[head] < 0:
                offsets[head] = np.random.randint(-head, length, size=1)[0]
        for tail in range(1, length):
            if (n - tail) + offsets[-tail] >= n:
                offsets[-tail] = np.random.randint(-length, tail, size=1)[0]
        return np.arange(n) + offsets

    def randomise_pairs(self, length=3, seed=None):
        """Re-pair modality 0 (images and masks) with a nearby slice of the same volume; modality 1 is untouched."""
        if seed is not None:
            np.random.seed(seed)
        rows = []
        for vol in self.volumes():
            own = np.nonzero(self.index == vol)[0]
            rows.append(own[self.random_pair_index(len(own), length)])
        rows = np.concatenate(rows)
        self.image_dict[0], self.masks_dict[0] = self.image_dict[0][rows], self.masks_dict[0][rows]
