"""Oracle model graphs: the reference's components restated over a flat
parameter dict (name -> torch tensor, Keras weight layouts: conv HWIO, dense
[in, out]).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Parameter prefixes (the product's components export the same names):
  EA0/ EA1/  per-modality UNet down path (DAFNet) or whole UNet (MMSDNet)
  EAS/       DAFNet shared bottleneck + up path + conv_anatomy
  EM/ SEG/ DEC/ FUS/ DM/ DI1/ DI2/ BAL/
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops as O


# ----------------------------------------------------------------------------
# initialisers (Keras 2.1.6 defaults (dd))
# ----------------------------------------------------------------------------
def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2
    return out * std


def init_kernel(rng, shape, kind):
    fi, fo = _fans(shape)
    if kind == 'he_normal':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / fi))
    if kind == 'glorot_normal':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / (fi + fo)))
    if kind == 'glorot_uniform':
        lim = math.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape)
    if kind == 'zeros':
        return np.zeros(shape)
    raise ValueError(kind)


class ParamBuilder(object):
    def __init__(self, seed, dtype=torch.float32):
        self.rng = np.random.RandomState(seed)
        self.P = OrderedDict()
        self.dtype = dtype

    def conv(self, name, k, cin, cout, init='glorot_uniform', bias=True):
        self.P[name + '/kernel'] = torch.tensor(init_kernel(self.rng, (k, k, cin, cout), init), dtype=self.dtype)
        if bias:
            self.P[name + '/bias'] = torch.zeros(cout, dtype=self.dtype)

    def dense(self, name, cin, cout, init='glorot_uniform'):
        self.P[name + '/kernel'] = torch.tensor(init_kernel(self.rng, (cin, cout), init), dtype=self.dtype)
        self.P[name + '/bias'] = torch.zeros(cout, dtype=self.dtype)

    def norm(self, name, c, norm='batch'):
        """utils/model_utils.normalise (model_utils.py:6-12): BatchNormalization, keras_contrib InstanceNormalization()
        (axis=None -> scalar gamma / beta) or the identity"""
        if norm == 'batch':
            self.bn(name, c)
        elif norm == 'instance':
            self.P[name + '/gamma'] = torch.ones(1, dtype=self.dtype)
            self.P[name + '/beta'] = torch.zeros(1, dtype=self.dtype)

    def bn(self, name, c):
        self.P[name + '/gamma'] = torch.ones(c, dtype=self.dtype)
        self.P[name + '/beta'] = torch.zeros(c, dtype=self.dtype)
        self.P[name + '/moving_mean'] = torch.zeros(c, dtype=self.dtype)
        self.P[name + '/moving_variance'] = torch.ones(c, dtype=self.dtype)


# ----------------------------------------------------------------------------
# UNet (models/unet.py:37-101, utils/model_utils.py:15-22)
# ----------------------------------------------------------------------------
def _build_conv_block(pb, name, cin, f, norm='batch'):
    pb.conv(name + 'a', 3, cin, f, 'he_normal'); pb.norm(name + 'a_bn', f, norm)
    pb.conv(name + 'b', 3, f, f, 'he_normal'); pb.norm(name + 'b_bn', f, norm)


def _normalise(l, P, name, training, upd):
    """the layer utils/model_utils.normalise built, told apart by the parameters it owns: 4 -> BatchNormalization, scalar
    gamma/beta -> keras_contrib InstanceNormalization() over (H, W, C) jointly with (x - mean) / (std + eps), none -> identity"""
    if name + '/moving_mean' in P:
        return O.batchnorm(l, P, name, training, upd)
    if name + '/gamma' in P:
        return O.instance_norm(l) * P[name + '/gamma'] + P[name + '/beta']
    return l


def _conv_block(x, P, name, training, upd):
    """models/unet.py:94-101."""
    l = O.conv2d(x, P[name + 'a/kernel'], P[name + 'a/bias'])
    l = torch.relu(_normalise(l, P, name + 'a_bn', training, upd))
    l = O.conv2d(l, P[name + 'b/kernel'], P[name + 'b/bias'])
    return torch.relu(_normalise(l, P, name + 'b_bn', training, upd))


def build_unet_down(pb, prefix, cin, f, norm='batch'):
    c = cin
    for i in range(4):
        _build_conv_block(pb, '%sd%d' % (prefix, i), c, f * 2 ** i, norm)
        c = f * 2 ** i


def build_unet_up(pb, prefix, f, out_channels, norm='batch'):
    _build_conv_block(pb, prefix + 'bott', f * 8, f * 16, norm)
    c = f * 16
    for i in (3, 2, 1, 0):
        fo = f * 2 ** i
        pb.conv('%su%d' % (prefix, i), 3, c, fo, 'he_normal'); pb.norm('%su%d_bn' % (prefix, i), fo, norm)
        _build_conv_block(pb, '%su%dc' % (prefix, i), 2 * fo, fo, norm)
        c = fo
    pb.conv(prefix + 'conv_anatomy', 1, f, out_channels)


def unet_down(x, P, prefix, training, upd):
    """models/unet.py:37-52: returns pooled output and the 4 skip tensors."""
    skips = []
    l = x
    for i in range(4):
        d = _conv_block(l, P, '%sd%d' % (prefix, i), training, upd)
        skips.append(d)
        l = O.maxpool2(d)
    return l, skips


def unet_up(l, skips, P, prefix, training, upd, rounding=True, return_presoftmax=False):
    """unet_bottleneck + unet_upsample + conv_anatomy softmax + Rounding
    (models/unet.py:54-86; model_components/anatomy_encoder.py:23-25,75-102).
    The up-conv has BN but a *linear* activation (unet.py:67,72,77,82)."""
    l = _conv_block(l, P, prefix + 'bott', training, upd)
    for i in (3, 2, 1, 0):
        n = '%su%d' % (prefix, i)
        l = O.conv2d(O.upsample2(l), P[n + '/kernel'], P[n + '/bias'])
        l = _normalise(l, P, n + '_bn', training, upd)
        l = torch.cat([l, skips[i]], dim=-1)
        l = _conv_block(l, P, n + 'c', training, upd)
    logits = O.conv2d(l, P[prefix + 'conv_anatomy/kernel'], P[prefix + 'conv_anatomy/bias'])
    soft = torch.softmax(logits, dim=-1)
    if return_presoftmax:
        return soft
    return O.round_ste(soft) if rounding else soft


def anatomy_encoder_dafnet(x, P, mod, training, upd, soft_only=False):
    """AnatomyEncoders.build (anatomy_encoder.py:37-73): per-modality down path,
    shared bottleneck/up path/conv_anatomy."""
    l, skips = unet_down(x, P, 'EA%d/' % mod, training, upd)
    return unet_up(l, skips, P, 'EAS/', training, upd, return_presoftmax=soft_only)


def anatomy_encoder_mmsdnet(x, P, mod, training, upd, soft_only=False):
    """anatomy_encoder.build (anatomy_encoder.py:13-30): a full UNet per modality."""
    l, skips = unet_down(x, P, 'EA%d/' % mod, training, upd)
    return unet_up(l, skips, P, 'EA%d/' % mod, training, upd, return_presoftmax=soft_only)


# ----------------------------------------------------------------------------
# modality encoder (model_components/modality_encoder.py:13-52)
# ----------------------------------------------------------------------------
def _valid_out(n, k, s):
    return (n - k) // s + 1


def build_modality_encoder(pb, H, W, s_ch=8, num_z=8):
    c = s_ch + 1
    h, w = H, W
    for i, f in enumerate((16, 32, 64, 128)):
        pb.conv('EM/c%d' % i, 3, c, f, 'he_normal')
        c = f
        h, w = _valid_out(h, 3, 2), _valid_out(w, 3, 2)
    pb.dense('EM/d0', h * w * c, 32, 'he_normal')
    pb.dense('EM/z_mean', 32, num_z)
    pb.dense('EM/z_log_var', 32, num_z)


def modality_encoder_mu(s, x, P):
    l = torch.cat([s, x], dim=-1)
    for i in range(4):
        l = O.leaky_relu(O.conv2d(l, P['EM/c%d/kernel' % i], P['EM/c%d/bias' % i], stride=2, padding='valid'))
    l = O.leaky_relu(O.dense(O.flatten(l), P['EM/d0/kernel'], P['EM/d0/bias']))
    z_mean = O.dense(l, P['EM/z_mean/kernel'], P['EM/z_mean/bias'])
    z_log_var = O.dense(l, P['EM/z_log_var/kernel'], P['EM/z_log_var/bias'])
    return z_mean, z_log_var


def modality_encoder(s, x, eps, P):
    """-> (z, kl) ; Enc_Modality_mu is modality_encoder_mu(...)[0] (dafnet.py:126)."""
    z_mean, z_log_var = modality_encoder_mu(s, x, P)
    return O.sampling(z_mean, z_log_var, eps), O.kl(z_mean, z_log_var)


# ----------------------------------------------------------------------------
# segmentor (model_components/segmentor.py:9-29)
# ----------------------------------------------------------------------------
def build_segmentor(pb, s_ch=8, num_masks=4):
    pb.conv('SEG/c0', 3, s_ch, 64, 'he_normal'); pb.bn('SEG/c0_bn', 64)
    pb.conv('SEG/c1', 3, 64, 64, 'he_normal'); pb.bn('SEG/c1_bn', 64)
    pb.conv('SEG/out', 1, 64, num_masks + 1)


def segmentor(s, P, training, upd, return_logits=False):
    l = torch.relu(O.batchnorm(O.conv2d(s, P['SEG/c0/kernel'], P['SEG/c0/bias']), P, 'SEG/c0_bn', training, upd))
    l = torch.relu(O.batchnorm(O.conv2d(l, P['SEG/c1/kernel'], P['SEG/c1/bias']), P, 'SEG/c1_bn', training, upd))
    logits = O.conv2d(l, P['SEG/out/kernel'], P['SEG/out/bias'])
    return logits if return_logits else torch.softmax(logits, dim=-1)


# ----------------------------------------------------------------------------
# decoders (model_components/decoder.py)
# ----------------------------------------------------------------------------
def build_decoder_film(pb, s_ch=8, num_z=8):
    pb.conv('DEC/c0', 3, s_ch, 8)
    for i in range(4):
        pb.conv('DEC/f%d_c1' % i, 3, 8, 8)
        pb.conv('DEC/f%d_c2' % i, 3, 8, 8)
        pb.dense('DEC/f%d_gamma' % i, num_z, 8)
        pb.dense('DEC/f%d_beta' % i, num_z, 8)
    pb.conv('DEC/out', 1, 8, 1, 'glorot_normal')


def decoder_film(s, z, P, return_pre_tanh=False):
    """_film_decoder/_film_layer/_gamma_beta_pred (decoder.py:36-64) + head (28)."""
    l = O.leaky_relu(O.conv2d(s, P['DEC/c0/kernel'], P['DEC/c0/bias']))
    for i in range(4):
        n = 'DEC/f%d' % i
        l1 = O.leaky_relu(O.conv2d(l, P[n + '_c1/kernel'], P[n + '_c1/bias']))
        l2 = O.conv2d(l1, P[n + '_c2/kernel'], P[n + '_c2/bias'])
        gamma = O.leaky_relu(O.dense(z, P[n + '_gamma/kernel'], P[n + '_gamma/bias']))
        beta = O.leaky_relu(O.dense(z, P[n + '_beta/kernel'], P[n + '_beta/bias']))
        l2 = O.leaky_relu(O.film(l2, gamma, beta))
        l = l1 + l2
    pre = O.conv2d(l, P['DEC/out/kernel'], P['DEC/out/bias'])
    return pre if return_pre_tanh else torch.tanh(pre)


SPADE_BLOCKS = ((128, 128), (128, 128), (128, 128), (128, 64), (64, 32), (32, 16))


def _build_spade(pb, name, f, s_ch):
    pb.conv(name + '_shared', 3, s_ch, 128)
    pb.conv(name + '_gamma', 3, 128, f)
    pb.conv(name + '_beta', 3, 128, f)


def build_decoder_spade(pb, H, W, s_ch=8, num_z=8):
    pb.dense('DEC/fc', num_z, H * W * 128 // 1024)
    for i, (fin, fout) in enumerate(SPADE_BLOCKS):
        n = 'DEC/b%d' % i
        fmid = min(fin, fout)
        _build_spade(pb, n + '_s0', fin, s_ch); pb.conv(n + '_c0', 3, fin, fmid)
        _build_spade(pb, n + '_s1', fmid, s_ch); pb.conv(n + '_c1', 3, fmid, fout)
        if fin != fout:
            _build_spade(pb, n + '_ss', fin, s_ch); pb.conv(n + '_cs', 1, fin, fout, bias=False)
    pb.conv('DEC/out', 1, 16, 1, 'glorot_normal')


def _spade(s, layer, P, name):
    """layers/spade.py:26-33."""
    layer = O.instance_norm(layer)
    a = O.resize_nearest(s, layer.shape[1], layer.shape[2])
    a = torch.relu(O.conv2d(a, P[name + '_shared/kernel'], P[name + '_shared/bias']))
    gamma = O.conv2d(a, P[name + '_gamma/kernel'], P[name + '_gamma/bias'])
    beta = O.conv2d(a, P[name + '_beta/kernel'], P[name + '_beta/bias'])
    return O.spade_cond(layer, gamma, beta)


def _spade_block(s, layer, P, n, fin, fout):
    """layers/spade.py:7-23."""
    l1 = _spade(s, layer, P, n + '_s0')
    l3 = O.conv2d(O.leaky_relu(l1, 0.2), P[n + '_c0/kernel'], P[n + '_c0/bias'])
    l4 = _spade(s, l3, P, n + '_s1')
    l6 = O.conv2d(O.leaky_relu(l4, 0.2), P[n + '_c1/kernel'], P[n + '_c1/bias'])
    if fin != fout:
        layer = _spade(s, layer, P, n + '_ss')
        layer = O.conv2d(layer, P[n + '_cs/kernel'], None)
    return layer + l6


def decoder_spade(s, z, P, return_pre_tanh=False):
    """_spade_decoder (decoder.py:67-81) + head (28)."""
    B, H, W, _ = s.shape
    l = O.dense(z, P['DEC/fc/kernel'], P['DEC/fc/bias']).reshape(B, H // 32, W // 32, 128)
    for i, (fin, fout) in enumerate(SPADE_BLOCKS):
        if i > 0:
            l = O.upsample2(l)
        l = _spade_block(s, l, P, 'DEC/b%d' % i, fin, fout)
    pre = O.conv2d(l, P['DEC/out/kernel'], P['DEC/out/bias'])
    return pre if return_pre_tanh else torch.tanh(pre)


# ----------------------------------------------------------------------------
# anatomy fuser (model_components/anatomy_fuser.py, layers/stn_spline.py:94-120)
# ----------------------------------------------------------------------------
def build_fuser(pb, H, W, s_ch=8):
    c = 2 * s_ch
    h, w = H, W
    for i in range(3):
        pb.conv('FUS/c%d' % i, 5, c, 20)
        c = 20
        h, w = h - 4, w - 4
        if i < 2:
            h, w = h // 2, w // 2
    pb.dense('FUS/d0', h * w * 20, 100)
    pb.dense('FUS/theta', 100, 50, 'zeros')


def locnet(a1, a2, P):
    l = torch.cat([a1, a2], dim=-1)
    for i in range(3):
        l = O.leaky_relu(O.conv2d(l, P['FUS/c%d/kernel' % i], P['FUS/c%d/bias' % i], padding='valid'))
        if i < 2:
            l = O.maxpool2(l)
    l = torch.tanh(O.dense(O.flatten(l), P['FUS/d0/kernel'], P['FUS/d0/bias']))
    theta = O.dense(l, P['FUS/theta/kernel'], P['FUS/theta/bias'])
    return theta.reshape(-1, 25, 2)


def anatomy_fuser(a1, a2, P):
    """-> (a1 deformed onto a2, max(a1_deformed, a2))  (anatomy_fuser.py:28-35)."""
    theta = locnet(a1, a2, P)
    a1_def = O.tps_warp(a1, theta)
    # keras Maximum = tf.maximum: value max, gradient ties go to the FIRST argument (dd)
    return a1_def, torch.where(a1_def >= a2, a1_def, a2)


# ----------------------------------------------------------------------------
# discriminator (models/discriminator.py:16-41)
# ----------------------------------------------------------------------------
def build_discriminator(pb, prefix, H, W, cin, f):
    pb.conv(prefix + 'c0', 4, cin, f, 'he_normal')
    h, w = _valid_out(H, 4, 2), _valid_out(W, 4, 2)
    c = f
    for i in range(3):
        s = 1 if i == 2 else 2
        pb.conv(prefix + 'c%d' % (i + 1), 4, c, f * 2 * 2 ** i, 'he_normal')
        # Spectral(dim) keeps u0 ~ U(-1, 1) drawn once at construction (spectralnorm.py:213)
        pb.P[prefix + 'c%d/u0' % (i + 1)] = torch.tensor(pb.rng.random_sample((c * 16, 1)) * 2 - 1., dtype=pb.dtype)
        c = f * 2 * 2 ** i
        h, w = _valid_out(h, 4, s), _valid_out(w, 4, s)
    pb.dense(prefix + 'out', h * w * c, 1)


def discriminator(x, P, prefix):
    l = O.leaky_relu(O.conv2d(x, P[prefix + 'c0/kernel'], P[prefix + 'c0/bias'], stride=2, padding='valid'), 0.2)
    for i in range(3):
        s = 1 if i == 2 else 2
        n = prefix + 'c%d' % (i + 1)
        l = O.leaky_relu(O.conv2d(l, P[n + '/kernel'], P[n + '/bias'], stride=s, padding='valid'), 0.2)
    return O.dense(O.flatten(l), P[prefix + 'out/kernel'], P[prefix + 'out/bias'])


def discriminator_reg(P, prefix):
    """Sum of the three Spectral regularisers (discriminator.py:39-40)."""
    tot = 0.
    for i in range(3):
        n = prefix + 'c%d' % (i + 1)
        tot = tot + O.spectral_reg(P[n + '/kernel'], P[n + '/u0'], 10.0)
    return tot


# ----------------------------------------------------------------------------
# balancer (model_components/balancer.py:11-38)
# ----------------------------------------------------------------------------
def build_balancer(pb, n_pairs=3):
    pb.dense('BAL/d0', 3, 5)
    pb.dense('BAL/beta', 5, n_pairs)


def balancer(x1, x2, x3, x4, P):
    def dice(a, b):
        inter = (a * b).sum(dim=(1, 2, 3))
        union = a.sum(dim=(1, 2, 3)) + b.sum(dim=(1, 2, 3))
        return ((2 * inter + 1e-12) / (union + 1e-12))[:, None]
    x = torch.cat([dice(x1, x) for x in (x2, x3, x4)], dim=1)
    l = torch.relu(O.dense(x, P['BAL/d0/kernel'], P['BAL/d0/bias']))
    return torch.softmax(O.dense(l, P['BAL/beta/kernel'], P['BAL/beta/bias']), dim=-1)


# ----------------------------------------------------------------------------
# whole-model parameter sets
# ----------------------------------------------------------------------------
def build_dafnet_params(seed, H, W, decoder_type='film', f=64, d_filters=64, num_masks=4, dtype=torch.float32):
    pb = ParamBuilder(seed, dtype)
    build_discriminator(pb, 'DM/', H, W, num_masks, d_filters)
    build_discriminator(pb, 'DI1/', H, W, 1, d_filters)
    build_discriminator(pb, 'DI2/', H, W, 1, d_filters)
    build_unet_down(pb, 'EA0/', 1, f)
    build_unet_down(pb, 'EA1/', 1, f)
    build_unet_up(pb, 'EAS/', f, 8)
    build_fuser(pb, H, W)
    build_modality_encoder(pb, H, W)
    build_segmentor(pb, 8, num_masks)
    if decoder_type == 'film':
        build_decoder_film(pb)
    else:
        build_decoder_spade(pb, H, W)
    build_balancer(pb)
    return pb.P


def build_mmsdnet_params(seed, H, W, decoder_type='film', f=64, d_filters=4, num_masks=4, dtype=torch.float32, num_mod=2):
    pb = ParamBuilder(seed, dtype)
    build_discriminator(pb, 'DM/', H, W, num_masks, d_filters)
    for m in range(num_mod):
        build_unet_down(pb, 'EA%d/' % m, 1, f)
        build_unet_up(pb, 'EA%d/' % m, f, 8)
    build_fuser(pb, H, W)
    build_modality_encoder(pb, H, W)
    build_segmentor(pb, 8, num_masks)
    if decoder_type == 'film':
        build_decoder_film(pb)
    else:
        build_decoder_spade(pb, H, W)
    return pb.P


NON_TRAINABLE_SUFFIXES = ('/moving_mean', '/moving_variance', '/u0')


def trainable_names(P, prefixes):
    return [k for k in P if k.startswith(tuple(prefixes)) and not k.endswith(NON_TRAINABLE_SUFFIXES)]
