"""Balancer: similarity weights between anatomies from their Dice overlap (reference model_components/balancer.py).
Built by DAFNet.build_generators (dafnet.py:129) but only USED by the automated-pairing graphs, which are a "next"
row; the model exists so that the checkpoint layout and the SWA list are complete."""
import logging

from .. import nn
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
        raise NotImplementedError('Balancer forward belongs to the automated-pairing graph (SURVEY 8f rank 4)')


def build(conf, rng=None):
    m = Balancer(conf, rng or global_rng())
    log.info('Balancer')
    return m
