"""Balancer: similarity weights between anatomies from their Dice overlap (reference model_components/balancer.py:11-38):
Dice of x1 against x2, x3, x4 -> Dense(5, relu) -> Dense(n_pairs) 'beta' -> softmax.  Built by DAFNet.build_generators
(dafnet.py:129); used by the automated-pairing graphs (dafnet.py:286-290,352-361) and by validation
(dafnet_executor.py:356-367).  Four inputs are hard-wired in the reference, hence n_pairs = 3."""
import logging

from .. import nn, ops
from ..utils.rng import global_rng

log = logging.getLogger('pair_selector')


class Balancer(nn.Model):
    def __init__(self, conf, rng):
        super(Balancer, self).__init__('Balancer')
        n_pairs = conf.n_pairs if hasattr(conf, 'n_pairs') else 1
        nn.dense_params(self, 'd0', 3, 5)
        nn.dense_params(self, 'beta', 5, n_pairs)
        self.finalize(rng)

    def forward(self, x1, x2, x3, x4, training=False):
        overlap = ops.overlap_dice(x1, [x2, x3, x4])                  # [B, 3]
        l = nn.dense(self, 'd0', overlap, act='relu')
        return ops.softmax(nn.dense(self, 'beta', l))


def build(conf, rng=None):
    m = Balancer(conf, rng or global_rng())
    log.info('Balancer')
    return m
