"""Decoders: (anatomy, z) -> image, FiLM or SPADE conditioned (reference model_components/decoder.py:12-81).

FiLM decoder (decoder.py:36-64): conv3x3 8->8 + LeakyReLU(0.3); 4 x _film_layer; conv1x1 -> tanh (glorot_normal).
_film_layer: l1 = LReLU(conv(x)); l2 = conv(l1); gamma, beta = LReLU(Dense8(z)); out = l1 + LReLU(l2*gamma+beta).
FiLM + LeakyReLU + residual Add are one kernel (csrc/pointwise.hip film_fwd).
"""
import logging

from .. import nn, ops
from ..layers import spade
from ..utils.rng import global_rng

log = logging.getLogger('decoder')


class Decoder(nn.Model):
    def __init__(self, conf, rng):
        super(Decoder, self).__init__('Decoder')
        self.conf = conf
        self.decoder_type = conf.decoder_type
        sc = conf.anatomy_encoder.output_shape[-1]
        if conf.decoder_type == 'film':
            nn.conv_params(self, 'c0', 3, sc, 8)
            for i in range(4):
                nn.conv_params(self, 'f%d_c1' % i, 3, 8, 8)
                nn.conv_params(self, 'f%d_c2' % i, 3, 8, 8)
                nn.dense_params(self, 'f%d_gamma' % i, conf.num_z, 8)
                nn.dense_params(self, 'f%d_beta' % i, conf.num_z, 8)
            last = 8
        elif conf.decoder_type == 'spade':
            last = spade.declare_spade_decoder(self, conf)
        else:
            raise ValueError('Unknown decoder_type value: ' + str(conf.decoder_type))
        nn.conv_params(self, 'out', 1, last, 1, 'glorot_normal')
        self.finalize(rng)
        self.output_shape = (None,) + tuple(conf.input_shape)

    def _film_decoder(self, s, z):
        l = nn.conv(self, 'c0', s, act='leaky', alpha=0.3)
        zs = ops.Shared(z, 8)          # z feeds the gamma and beta layers of all four FiLM layers
        for i in range(4):
            n = 'f%d' % i
            l1 = ops.Shared(nn.conv(self, n + '_c1', l, act='leaky', alpha=0.3), 2)      # second convolution + residual
            l2 = nn.conv(self, n + '_c2', l1.use())
            gamma = nn.dense(self, n + '_gamma', zs.use(), act='leaky', alpha=0.3)
            beta = nn.dense(self, n + '_beta', zs.use(), act='leaky', alpha=0.3)
            l = ops.film(l2, gamma, beta, res=l1.use(), alpha=0.3)
        return l

    def forward(self, s, z, training=False):
        if self.decoder_type == 'film':
            l = self._film_decoder(s, z)
        else:
            l = spade.spade_decoder(self, self.conf, s, z)
        return nn.conv(self, 'out', l, act='tanh')


def build(conf, rng=None):
    """Build a decoder that generates an image by combining an anatomical and a modality representation."""
    model = Decoder(conf, rng or global_rng())
    log.info('Decoder')
    model.summary(print_fn=log.debug)
    return model
