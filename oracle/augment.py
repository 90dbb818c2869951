"""ORACLE (test infrastructure only; PARITY UNPINNED -- Keras 2.1.6 is not importable here, restated from its published
source): keras.preprocessing.image.ImageDataGenerator(rotation_range=r).flow(x, batch_size, seed) as used by reference
model_executors/base_executor.py:37-78,103-110, on the host with scipy -- the same scipy call keras makes.

    Iterator._flow_index (keras 2.1.6 preprocessing/image.py): per batch np.random.seed(seed + total_batches_seen);
        a new permutation when batch_index == 0; rows = order[cur : cur + B]
    random_transform: theta = deg2rad(uniform(-r, r)); rotation about (h/2 + 0.5, w/2 + 0.5)
        (transform_matrix_offset_center); apply_transform -> ndi.affine_transform(channel, M[:2,:2], M[:2,2], order=1,
        mode='nearest', cval=0.) per channel.
"""
import numpy as np
from scipy import ndimage as ndi


def transform_matrix(theta, h, w):
    rot = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
    ox, oy = float(h) / 2 + 0.5, float(w) / 2 + 0.5
    offset = np.array([[1, 0, ox], [0, 1, oy], [0, 0, 1]])
    reset = np.array([[1, 0, -ox], [0, 1, -oy], [0, 0, 1]])
    return offset @ rot @ reset


def apply_transform(x, m, order=1):
    """x [H,W,C]"""
    chans = [ndi.affine_transform(x[..., c], m[:2, :2], m[:2, 2], order=order, mode='nearest', cval=0.)
             for c in range(x.shape[-1])]
    return np.stack(chans, axis=-1)


class KerasFlowOracle(object):
    """One keras NumpyArrayIterator(shuffle=True) with rotation only.  Uses (and reseeds) the GLOBAL numpy RNG."""

    def __init__(self, x, batch_size, seed, rotation_range=20., order=1):
        self.x, self.batch_size, self.seed, self.rot, self.order = x, batch_size, seed, rotation_range, order
        self.n = x.shape[0]
        self.batch_index = 0
        self.total_batches_seen = 0
        self.index_array = None

    def __next__(self):
        if self.seed is not None:
            np.random.seed(self.seed + self.total_batches_seen)
        if self.batch_index == 0:
            self.index_array = np.random.permutation(self.n)
        cur = (self.batch_index * self.batch_size) % self.n
        if self.n > cur + self.batch_size:
            self.batch_index += 1
        else:
            self.batch_index = 0
        self.total_batches_seen += 1
        rows = self.index_array[cur:cur + self.batch_size]
        out = np.zeros((len(rows),) + self.x.shape[1:], np.float32)
        for i, j in enumerate(rows):
            x = self.x[j].astype(np.float32)
            theta = np.deg2rad(np.random.uniform(-self.rot, self.rot)) if self.rot else 0
            out[i] = apply_transform(x, transform_matrix(theta, x.shape[0], x.shape[1]), self.order)
        return out

    next = __next__
