"""reference utils/distributions.py:4-11"""
import numpy as np


class NormalDistribution(object):
    def __init__(self):
        self.mu = 0
        self.sigma = 1

    def sample(self, N):
        return np.random.normal(self.mu, self.sigma, N)
