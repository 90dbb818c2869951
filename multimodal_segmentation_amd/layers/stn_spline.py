"""Thin-plate-spline spatial transformer (reference layers/stn_spline.py + layers/interpolate_spline.py).

The reference fits a polyharmonic spline (order 2) per sample with a 28x28 `matrix_solve` and evaluates it at all
H*W grid points.  With inverse=False (anatomy_fuser.py:30) the spline's centres are the FIXED 5x5 control grid, so
the interpolant is linear in the values:  loc = grid + Mb @ theta  with a constant Mb [H*W, 25].  Mb is computed once
here on the host in float64 from the same linear system (interpolate_spline.py:76-179) and uploaded; the per-step
work is csrc/tps.hip (25-term dot product + 4-tap gather per pixel, backward dtheta = Mb^T dloc).
"""
import numpy as np
import torch

from .. import nn, ops


def nDgrid(dims, normalise=True):
    """reference stn_spline.py:70-91 (numpy, float64): rows (row/(H-1), col/(W-1))"""
    grid = np.mgrid[:dims[0], :dims[1]].reshape((2, -1)).T.astype(np.float64)
    if normalise:
        grid = grid / (np.array(dims, np.float64)[None] - 1)
    return grid


def _phi(r):
    return 0.5 * r * np.log(np.maximum(r, 1e-10))       # interpolate_spline.py:198-199 (order 2)


def _sq_dist(x, y):
    return (x * x).sum(1)[:, None] - 2 * x @ y.T + (y * y).sum(1)[None]   # interpolate_spline.py:30-51


def tps_basis(vol_shape, cp_dims):
    """Mb [H*W, n_cp] float64: interpolate_spline(train_points=cp, train_values=I, query_points=grid)."""
    c = nDgrid(cp_dims)
    q = nDgrid(vol_shape)
    n, d = c.shape
    A = _phi(_sq_dist(c, c))
    Bm = np.concatenate([c, np.ones((n, 1))], 1)
    lhs = np.block([[A, Bm], [Bm.T, np.zeros((d + 1, d + 1))]])
    rhs = np.concatenate([np.eye(n), np.zeros((d + 1, n))], 0)
    wv = np.linalg.solve(lhs, rhs)
    w, v = wv[:n], wv[n:]
    return _phi(_sq_dist(q, c)) @ w + np.concatenate([q, np.ones((q.shape[0], 1))], 1) @ v


class ThinPlateSpline2D(object):
    """Keras-layer-shaped wrapper: tps([vol, cp_offsets]) -> warped vol (stn_spline.py:14-67)."""

    def __init__(self, input_volume_shape, cp_dims, num_channels, inverse=False, order=2):
        if inverse or order != 2:
            raise NotImplementedError('only inverse=False, order=2 (what anatomy_fuser.py:30 uses)')
        self.vol_shape = tuple(input_volume_shape)
        self.cp_dims = tuple(cp_dims)
        self.num_channels = num_channels
        self._Mb_host = tps_basis(self.vol_shape, self.cp_dims).astype(np.float32)
        self._Mb = {}

    def basis(self, device):
        m = self._Mb.get(device)
        if m is None:
            m = torch.from_numpy(self._Mb_host).to(device)
            self._Mb[device] = m
        return m

    def __call__(self, args):
        vol, cp_offsets = args
        return ops.tps_warp(vol, cp_offsets, self.basis(vol.device))


def declare_locnet(m, input_shape1, input_shape2, output_shape):
    """reference build_locnet (stn_spline.py:94-120)"""
    c = input_shape1[-1] + input_shape2[-1]
    h, w = input_shape1[0], input_shape1[1]
    for i in range(3):
        nn.conv_params(m, 'c%d' % i, 5, c, 20)
        c = 20
        h, w = h - 4, w - 4
        if i < 2:
            h, w = h // 2, w // 2
    nn.dense_params(m, 'd0', h * w * 20, 100)
    nn.dense_params(m, 'theta', 100, output_shape, 'zeros')


def locnet(m, input1, input2):
    l = nn.conv(m, 'c0', input1, padding='valid', act='leaky', alpha=0.3, x2=input2)   # Concatenate folded in
    l = ops.maxpool2(l)
    l = nn.conv(m, 'c1', l, padding='valid', act='leaky', alpha=0.3)
    l = ops.maxpool2(l)
    l = nn.conv(m, 'c2', l, padding='valid', act='leaky', alpha=0.3)
    l = nn.dense(m, 'd0', l.reshape(l.shape[0], -1), act='tanh')
    theta = nn.dense(m, 'theta', l)
    return theta.reshape(theta.shape[0], -1, 2)
