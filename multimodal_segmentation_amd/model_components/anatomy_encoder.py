"""Anatomy encoders (reference model_components/anatomy_encoder.py).

build(conf, name)                 : one full UNet per modality (MMSDNet; anatomy_encoder.py:13-30)
AnatomyEncoders(mods).build(conf) : two per-modality down paths + ONE shared bottleneck / up path / conv_anatomy
                                    (DAFNet; anatomy_encoder.py:32-155)
Output: s = Rounding(softmax(conv1x1(features))) in {0,1}^{HxWx8}; the pre-rounding softmax is kept on the model
as `.last_soft` (used for teacher-forced parity checks across the rounding discontinuity).
"""
import logging

from .. import nn, ops
from ..models import unet
from ..utils.rng import global_rng

log = logging.getLogger('anatomy_encoder')


def _check(conf):
    if conf.normalise not in ('batch', 'instance', None, 'None', 'none'):
        raise ValueError("anatomy encoder: normalise must be 'batch', 'instance' or None (utils/model_utils.py:6-12)")
    if not (1 <= conf.downsample <= 4):
        raise ValueError('Unet downsample must be in 1..4')


class SharedDecoder(nn.Model):
    """AnatomyEncoders.build_decoder (anatomy_encoder.py:104-155): layers l0_1 .. l40 + the shared conv_anatomy."""

    def __init__(self, conf, rng, name='Enc_Anatomy_shared'):
        super(SharedDecoder, self).__init__(name)
        self.norm = _norm_of(conf)
        unet.declare_unet_up(self, conf.filters, conf.out_channels, conf.downsample, self.norm)
        self.finalize(rng)


def _norm_of(conf):
    """utils/model_utils.normalise: anything but 'instance' / 'batch' is the identity"""
    return conf.normalise if conf.normalise in ('batch', 'instance') else None


class AnatomyEncoder(nn.Model):
    """x [B,H,W,1] -> s [B,H,W,out_channels].  `shared`: a SharedDecoder whose weights are used (and reported by
    get_weights) instead of an own bottleneck/up path."""

    def __init__(self, conf, rng, name, shared=None):
        super(AnatomyEncoder, self).__init__(name)
        _check(conf)
        self.conf = conf
        self.norm = _norm_of(conf)
        unet.declare_unet_down(self, conf.input_shape[-1], conf.filters, conf.downsample, self.norm)
        if shared is None:
            unet.declare_unet_up(self, conf.filters, conf.out_channels, conf.downsample, self.norm)
            self.up = self
        else:
            self.up = shared
            self.shared = [shared]
        self.finalize(rng)
        self.input_shape = (None,) + tuple(conf.input_shape)
        self.output_shape = (None,) + tuple(conf.output_shape)
        self.last_soft = None

    def forward(self, x, training=False):
        ds = self.conf.downsample
        l, skips = unet.unet_downsample(self, x, training, ds, self.norm)
        l = unet.unet_bottleneck_upsample(self.up, l, skips, training, ds, self.norm)
        logits = nn.conv(self.up, 'conv_anatomy', l)
        soft, rounded = ops.softmax_round(logits)        # Conv2D(.., softmax) + Rounding (anatomy_encoder.py:23-25)
        self.last_soft = soft
        out = rounded if self.conf.rounding else soft
        if _rounding_hook[0] is not None:                # observers of the Rounding boundary (None in production; see set_rounding_hook)
            out = _rounding_hook[0](self, out)
        return out


_rounding_hook = [None]


def set_rounding_hook(fn):
    """fn(encoder, anatomy) -> anatomy, called with every anatomy factor an encoder returns; None removes it.  The Rounding layer
    makes everything downstream discontinuous in the encoder's logits, so comparisons against another implementation are made
    piecewise: a harness installs a hook that swaps in the other side's rounded anatomy with a straight-through gradient
    (tests/helpers.py::teacher_forcing).  The graphs themselves carry no such argument.  Returns the previous hook."""
    prev = _rounding_hook[0]
    _rounding_hook[0] = fn
    return prev


def build(conf, name='Enc_Anatomy', rng=None):
    """Build a UNet based encoder to extract anatomical information from the image."""
    model = AnatomyEncoder(conf, rng or global_rng(), name)
    log.info('Enc_Anatomy')
    model.summary(print_fn=log.debug)
    return model


class AnatomyEncoders(object):
    def __init__(self, modalities):
        self.modalities = modalities

    def build(self, conf, rng=None):
        rng = rng or global_rng()
        shared = SharedDecoder(conf, rng)
        return [AnatomyEncoder(conf, rng, 'Enc_Anatomy_%s' % mod, shared) for mod in self.modalities[:2]]
