"""Anatomy fuser: deform anatomy1 onto anatomy2 with a thin-plate-spline STN, then fuse with a pixel-wise max
(reference model_components/anatomy_fuser.py:12-38, layers/stn_spline.py)."""
import logging

from .. import nn, ops
from ..layers import stn_spline
from ..utils.rng import global_rng

log = logging.getLogger('anatomy_fuser')


class AnatomyFuser(nn.Model):
    def __init__(self, conf, rng):
        super(AnatomyFuser, self).__init__('Anatomy_Fuser')
        shp = conf.anatomy_encoder.output_shape
        self.cp = [5, 5]
        stn_spline.declare_locnet(self, shp, shp, self.cp[0] * self.cp[1] * 2)
        self.finalize(rng)
        self.tps = stn_spline.ThinPlateSpline2D(shp[:-1], self.cp, shp[-1])
        self.output_shape = [(None,) + tuple(shp)] * 2

    def forward(self, anatomy1, anatomy2, training=False):
        a1 = ops.Shared(anatomy1, 2)       # localisation network + the warp itself
        a2 = ops.Shared(anatomy2, 2)       # localisation network + the fusion
        theta = stn_spline.locnet(self, a1.use(), a2.use())
        d = ops.Shared(self.tps([a1.use(), theta]), 2)                # returned + fused
        anatomy1_deformed = d.use()
        anatomy_fused = ops.maximum(d.use(), a2.use())                # keras.layers.Maximum (anatomy_fuser.py:33)
        self.last_theta = theta
        return [anatomy1_deformed, anatomy_fused]


def build(conf, rng=None):
    model = AnatomyFuser(conf, rng or global_rng())
    log.info('Anatomy fuser')
    model.summary(print_fn=log.debug)
    return model
