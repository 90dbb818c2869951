"""UNet of 4 down-sampling and 4 up-sampling blocks (reference models/unet.py:37-101, utils/model_utils.py:6-22).

Each conv_block is (conv3x3 same, he_normal -> BatchNorm -> ReLU) x 2; down path ends each level with a 2x2 max
pool; the up path is nearest x2 -> conv3x3 -> BN -> *linear* -> concat(skip) -> conv_block.  On the MI355X the
up-sampling and the concatenation are folded into the im2col gather of the following convolution
(csrc/conv.hip), and bias / BN-apply / ReLU are fused, so neither UpSampling2D nor Concatenate touches HBM.

The functions below declare parameters on / apply layers of a nn.Model `m` so that the DAFNet encoders can put
the down path and the (shared) bottleneck + up path into different models (model_components/anatomy_encoder.py).
"""
from .. import nn


def declare_conv_block(m, name, cin, f, norm='batch'):
    nn.conv_params(m, name + 'a', 3, cin, f, 'he_normal'); nn.norm_params(m, name + 'a_bn', f, norm)
    nn.conv_params(m, name + 'b', 3, f, f, 'he_normal'); nn.norm_params(m, name + 'b_bn', f, norm)


def conv_block(m, name, x, training, x2=None, norm='batch'):
    """reference models/unet.py:94-101; `norm`: 'batch' | 'instance' | None (utils/model_utils.py:6-12)"""
    l = nn.conv_norm(m, name + 'a', name + 'a_bn', x, training, relu=True, x2=x2, norm=norm)
    return nn.conv_norm(m, name + 'b', name + 'b_bn', l, training, relu=True, norm=norm)


def declare_unet_down(m, cin, f, downsample=4, norm='batch'):
    c = cin
    for i in range(downsample):
        declare_conv_block(m, 'd%d' % i, c, f * 2 ** i, norm)
        c = f * 2 ** i


def unet_downsample(m, x, training, downsample=4, norm='batch'):
    """reference models/unet.py:37-52 -> (pooled tensor, [d_l0 .. d_l3])"""
    from .. import ops
    skips = []
    l = x
    for i in range(downsample):
        d = conv_block(m, 'd%d' % i, l, training, norm=norm)
        l, skip = ops.maxpool2_skip(d)        # d feeds the pooling and the skip concatenation: one node, gradients added in one pass
        skips.append(skip)
    return l, skips


def declare_unet_up(m, f, out_channels, downsample=4, norm='batch'):
    declare_conv_block(m, 'bott', f * 2 ** (downsample - 1), f * 2 ** downsample, norm)
    c = f * 2 ** downsample
    for i in reversed(range(downsample)):
        fo = f * 2 ** i
        nn.conv_params(m, 'u%d' % i, 3, c, fo, 'he_normal'); nn.norm_params(m, 'u%d_bn' % i, fo, norm)
        declare_conv_block(m, 'u%dc' % i, 2 * fo, fo, norm)
        c = fo
    nn.conv_params(m, 'conv_anatomy', 1, f, out_channels)


def unet_bottleneck_upsample(m, l, skips, training, downsample=4, norm='batch'):
    """reference models/unet.py:54-86 (bottleneck + up path); returns the f-channel feature map"""
    l = conv_block(m, 'bott', l, training, norm=norm)
    for i in reversed(range(downsample)):
        n = 'u%d' % i
        # UpSampling2D(2) + Conv2D fused; BatchNorm with activation='linear' (unet.py:67,72,77,82)
        l = nn.conv_norm(m, n, n + '_bn', l, training, relu=False, upsample=True, norm=norm)
        l = conv_block(m, n + 'c', l, training, x2=skips[i], norm=norm)  # Concatenate([l, skip]) fused into the conv
    return l
