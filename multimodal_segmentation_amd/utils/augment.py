"""Training-batch iterators with the reference's augmentation (model_executors/base_executor.py:37-78,103-110).

The reference wraps every training array in keras `ImageDataGenerator(rotation_range=20., <everything else off>)
.flow(array, batch_size=conf.batch_size, seed=conf.seed)` and zips the iterators, relying on the shared seed to apply
the SAME shuffle and the SAME rotation to images and masks.  Keras 2.1.6 (un-vendored; restated from its published
source, unverifiable here -- see DESIGN.md) does, for batch number k of one iterator:

    np.random.seed(seed + k)                              # the GLOBAL numpy RNG
    if this batch starts a pass: order = np.random.permutation(n)
    rows = order[i*B : (i+1)*B]                           # last batch of a pass may be short
    for each row: theta = deg2rad(np.random.uniform(-20, 20)); rotate about the centre with
                  scipy.ndimage.affine_transform(channel, R, offset, order=1, mode='nearest')

`RotationFlow` draws exactly that stream once per batch (every zipped keras iterator would redraw the identical numbers,
leaving the global RNG in the same state -- which matters, because the executors draw z samples and pool indices from
the global RNG right after `next(gen)`), keeps the arrays resident in HBM and produces the rotated batch with one
`mmseg_affine_gather` launch per array.
"""
import numpy as np
import torch

from .. import nn, ops


def rotation_matrices(thetas, H, W):
    """[B,6] fp32 rows (m0..m5): src_row = m0*r + m1*c + m2, src_col = m3*r + m4*c + m5 -- keras' rotation about
    (H/2 + 0.5, W/2 + 0.5) (transform_matrix_offset_center), composed on the host in fp64."""
    thetas = np.asarray(thetas, np.float64)
    cos, sin = np.cos(thetas), np.sin(thetas)
    oh, ow = H / 2.0 + 0.5, W / 2.0 + 0.5
    m = np.stack([cos, -sin, oh - cos * oh + sin * ow, sin, cos, ow - sin * oh - cos * ow], axis=1)
    return m.astype(np.float32)


class KerasFlowStream(object):
    """The (rows, thetas) stream of keras' NumpyArrayIterator(shuffle=True, seed=s) with only rotation enabled."""

    def __init__(self, n, batch_size, seed, rotation_range):
        self.n, self.batch_size, self.seed, self.rotation_range = int(n), int(batch_size), seed, float(rotation_range)
        self.total_batches_seen = 0
        self.batch_index = 0
        self.order = None

    def next(self):
        if self.seed is not None:
            np.random.seed(self.seed + self.total_batches_seen)
        if self.batch_index == 0:
            self.order = np.random.permutation(self.n)
        start = (self.batch_index * self.batch_size) % self.n
        self.batch_index = self.batch_index + 1 if self.n > start + self.batch_size else 0
        self.total_batches_seen += 1
        rows = self.order[start:start + self.batch_size]
        r = self.rotation_range
        thetas = [np.deg2rad(np.random.uniform(-r, r)) if r else 0.0 for _ in rows]
        return rows, np.asarray(thetas, np.float64)


class RotationFlow(object):
    """Iterator over aligned arrays [N,H,W,C_i] -> tuple of rotated device batches [B,H,W,C_i] (a single array yields a
    bare tensor, like the reference's single-generator case)."""

    def __init__(self, arrays, batch_size, seed, device, rotation_range=20., order=1):
        self.device = torch.device(device)
        self.arrays = [a if isinstance(a, torch.Tensor) else
                       torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device) for a in arrays]
        n = self.arrays[0].shape[0]
        assert all(a.shape[0] == n for a in self.arrays), [tuple(a.shape) for a in self.arrays]
        self.H, self.W = int(self.arrays[0].shape[1]), int(self.arrays[0].shape[2])
        self.stream = KerasFlowStream(n, batch_size, seed, rotation_range)
        self.order = order

    def __iter__(self):
        return self

    def __next__(self):
        rows, thetas = self.stream.next()
        rows_d = nn.host_to_device(rows, self.device, np.int32)
        mat_d = nn.host_to_device(rotation_matrices(thetas, self.H, self.W), self.device, np.float32)
        out = tuple(ops.affine_gather(a, rows_d, mat_d, self.order) for a in self.arrays)
        return out if len(out) > 1 else out[0]

    next = __next__
