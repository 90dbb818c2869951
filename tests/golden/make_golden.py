#!/usr/bin/env python
"""Generates tests/golden/reference_helpers.npz by importing the two modules of the reference that import in this
container (everything else needs Keras 2.1.6 / TF 1.4, which are absent): utils/data_utils.py and
utils/distributions.py.  Run once from the repo root with /root/reference present:

    python tests/golden/make_golden.py

The fixture holds inputs and outputs only (data), no reference source.  It pins the host-side random draws of the
training step: the fake-pool sampling order (data_utils.sample) and the z samples (NormalDistribution.sample) under a
fixed numpy seed.
"""
import os
import sys

import numpy as np

REF = '/root/reference'
sys.path.insert(0, REF)
from utils import data_utils            # noqa: E402
from utils.distributions import NormalDistribution  # noqa: E402

out = {}
pool = np.arange(24 * 3, dtype=np.float32).reshape(24, 3)
for seed in (0, 1, 1234):
    np.random.seed(seed)
    out['sample_seed%d' % seed] = data_utils.sample(pool, 8)
    out['normal_seed%d' % seed] = NormalDistribution().sample((4, 8))
out['sample_seedarg7'] = data_utils.sample(pool, 5, seed=7)
out['pool'] = pool
# crop / pad helpers used by the loaders (host side; "next" row), pinned for later rounds
img = np.arange(2 * 10 * 12 * 1, dtype=np.float32).reshape(2, 10, 12, 1)
out['crop_same_in'] = img
out['crop_same_out'] = data_utils.crop_same([img], [img], size=(8, 8))[0][0]
# more crop / pad / rescale / normalise cases for the data containers: odd differences (the 'equal'
# crop removes ceil(diff/2) on BOTH sides and is then padded back by one), constant and edge padding, left/right modes
rs = np.random.RandomState(5)
a = rs.rand(3, 11, 9, 2).astype(np.float32)
b = rs.rand(3, 11, 9, 1).astype(np.float32)
out['cs2_img'], out['cs2_msk'] = a, b
for tag, kw in (('odd_const', dict(size=(8, 6), pad_mode='constant')), ('pad_edge', dict(size=(14, 12), pad_mode='edge')),
                ('pad_const', dict(size=(14, 12), pad_mode='constant')), ('left', dict(size=(8, 6), mode='left')),
                ('right', dict(size=(8, 6), mode='right')), ('mixed', dict(size=(14, 6), pad_mode='constant'))):
    im, mk = data_utils.crop_same([a], [b], **kw)
    out['cs2_%s_img' % tag], out['cs2_%s_msk' % tag] = im[0], mk[0]
out['rescale_in'] = (rs.rand(2, 5, 5, 1) * 7 - 3)
out['rescale_out'] = data_utils.rescale(out['rescale_in'])
out['rescale_const_out'] = data_utils.rescale(np.full((2, 3, 3, 1), 4.0))
out['normalise_out'] = data_utils.normalise(out['rescale_in'])
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_helpers.npz'), **out)
print('wrote', sorted(out))
