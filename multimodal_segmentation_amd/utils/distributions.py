"""Prior of the modality factor z (interface of reference utils/distributions.py:4-11).

The executors draw the Z-regressor's targets with `NormalDistribution().sample((batch, num_z))`
(model_executors/dafnet_executor.py:497-499).  The draw must come from numpy's GLOBAL generator: the data iterators reseed
it before every batch (utils/augment.py), which makes the z samples of an iteration a function of the batch index exactly
as in the reference -- tests/golden/reference_helpers.npz pins the stream.  A private generator can be supplied for
experiments that want independent draws.
"""
import numpy as np


class NormalDistribution(object):
    """N(mu, sigma^2); `sample(shape)` -> float64 array of that shape."""

    def __init__(self, mu=0, sigma=1, generator=None):
        self.mu, self.sigma = mu, sigma
        self._gen = generator            # None: numpy's global RandomState (the reference's behaviour)

    def sample(self, N):
        draw = np.random.normal if self._gen is None else self._gen.normal
        return draw(self.mu, self.sigma, N)

    def __repr__(self):
        return 'NormalDistribution(mu=%r, sigma=%r)' % (self.mu, self.sigma)
