"""Segmentor: anatomy factor -> masks (+1 background channel) (reference model_components/segmentor.py:9-29)."""
import logging

from .. import nn, ops
from ..utils.rng import global_rng

log = logging.getLogger('segmentor')


class Segmentor(nn.Model):
    def __init__(self, conf, rng):
        super(Segmentor, self).__init__('Segmentor')
        cin = conf.anatomy_encoder.output_shape[-1]
        nn.conv_params(self, 'c0', 3, cin, 64, 'he_normal'); nn.bn_params(self, 'c0_bn', 64)
        nn.conv_params(self, 'c1', 3, 64, 64, 'he_normal'); nn.bn_params(self, 'c1_bn', 64)
        nn.conv_params(self, 'out', 1, 64, conf.num_masks + 1)      # +1 output for background
        self.finalize(rng)
        self.input_shape = (None,) + tuple(conf.anatomy_encoder.output_shape)
        self.output_shape = self.input_shape[:-1] + (conf.num_masks + 1,)

    def forward(self, s, training=False):
        l = nn.conv_bn(self, 'c0', 'c0_bn', s, training, relu=True)
        l = nn.conv_bn(self, 'c1', 'c1_bn', l, training, relu=True, y16=False)      # the 1x1 head (5 outputs) reads fp32
        self.last_logits = nn.conv(self, 'out', l)
        return ops.softmax(self.last_logits)


def build(conf, rng=None):
    """Build a segmentation network that converts anatomical maps to segmentation masks."""
    model = Segmentor(conf, rng or global_rng())
    log.info('Segmentor')
    model.summary(print_fn=log.debug)
    return model
