"""SPADE conditioning and the SPADE decoder body (reference layers/spade.py:7-58, model_components/decoder.py:67-81).

_spade: InstanceNorm(no affine) over (H,W,C) -> nearest-resize the anatomy to the layer's size -> conv3x3 8->128 ReLU
-> conv3x3 128->f (gamma), 128->f (beta) -> x*(1+gamma)+beta.  spade_block: SPADE -> LeakyReLU(0.2) -> conv3x3 ->
SPADE -> LeakyReLU(0.2) -> conv3x3 (+ SPADE -> conv1x1 no-bias shortcut when fin != fout) -> Add.
InstanceNorm + modulation + LeakyReLU are one fused kernel (csrc/norm.hip).
"""
from .. import nn, ops

SPADE_BLOCKS = ((128, 128), (128, 128), (128, 128), (128, 64), (64, 32), (32, 16))


def _declare_spade(m, name, f, s_ch):
    nn.conv_params(m, name + '_shared', 3, s_ch, 128)
    nn.conv_params(m, name + '_gamma', 3, 128, f)
    nn.conv_params(m, name + '_beta', 3, 128, f)


def declare_spade_decoder(m, conf):
    H, W = conf.input_shape[0], conf.input_shape[1]
    if H % 32 or W % 32:
        raise ValueError('SPADE decoder needs H, W multiples of 32 (decoder.py:68-69)')
    s_ch = conf.anatomy_encoder.output_shape[-1]
    nn.dense_params(m, 'fc', conf.num_z, H * W * 128 // 1024)
    for i, (fin, fout) in enumerate(SPADE_BLOCKS):
        n = 'b%d' % i
        fmid = min(fin, fout)
        _declare_spade(m, n + '_s0', fin, s_ch); nn.conv_params(m, n + '_c0', 3, fin, fmid)
        _declare_spade(m, n + '_s1', fmid, s_ch); nn.conv_params(m, n + '_c1', 3, fmid, fout)
        if fin != fout:
            _declare_spade(m, n + '_ss', fin, s_ch); nn.conv_params(m, n + '_cs', 1, fin, fout, bias=False)
    return SPADE_BLOCKS[-1][1]


def spade_hidden(m, name, a):
    """the unit's shared 3x3 convolution + ReLU (layers/spade.py:27).  Its 128-channel output is by far the largest tensor of the
    decoder (1.6 GB per unit in fp32 at 256 x 256 with the 48 decoder passes of an iteration batched); with conf.act_storage = 'half' it and its gradient live
    in HBM as 16-bit tensors (the gamma / beta convolutions and their gradients read / write them as such)."""
    import torch
    half = ops.act16_dtype()
    return nn.conv(m, name + '_shared', a, act='relu', out_dtype=half if half is not None else torch.float32)


def _spade(m, name, anatomy_input, layer, act_alpha):
    """layers/spade.py:26-33 (+ the LeakyReLU that follows it in spade_block when act_alpha >= 0)"""
    a = ops.resize_nearest_down(anatomy_input, layer.shape[1], layer.shape[2])
    a = spade_hidden(m, name, a)
    # gamma and beta: two Conv2D(f, 3) of the same 128-channel tensor -> one convolution with 2f output channels (round 3): the
    # hidden tensor is read once instead of twice, forward and backward; per output channel the same arithmetic
    # (with conf.act_storage = 'half' the fused tensor and its gradient live in HBM in the 16-bit type, like the hidden tensor)
    import torch
    half = ops.act16_dtype()
    gb = nn.conv_pair(m, name + '_gamma', name + '_beta', a, out_dtype=half if half is not None else torch.float32)
    # the modulated output feeds a convolution (c0 / c1 / cs): stored in 16 bits too when that convolution can read it (MFMA fast path:
    # input channels a multiple of 32); the unit's INPUT (a convolution output / the up-sampled trunk) stays fp32
    y_dt = half if (half is not None and layer.shape[-1] % 32 == 0) else torch.float32
    return ops.instnorm_spade_gb(layer, gb, act_alpha, out_dtype=y_dt)


def spade_block(m, n, anatomy_input, layer, fin, fout):
    """layers/spade.py:7-23"""
    layer = ops.Shared(layer, 2)                   # main branch + shortcut
    l2 = _spade(m, n + '_s0', anatomy_input.use(), layer.use(), 0.2)
    l3 = nn.conv(m, n + '_c0', l2)
    l5 = _spade(m, n + '_s1', anatomy_input.use(), l3, 0.2)
    l6 = nn.conv(m, n + '_c1', l5)
    if fin != fout:
        sc = _spade(m, n + '_ss', anatomy_input.use(), layer.use(), -1.0)
        sc = nn.conv(m, n + '_cs', sc)
    else:
        sc = layer.use()
    return ops.add(sc, l6)


def spade_decoder(m, conf, anatomy_input, modality_input):
    """decoder.py:67-81"""
    B = anatomy_input.shape[0]
    H, W = conf.input_shape[0], conf.input_shape[1]
    l = nn.dense(m, 'fc', modality_input).reshape(B, H // 32, W // 32, 128)
    # every SPADE unit conditions on the anatomy (15 units): one alias per unit, their gradients are added by two launches
    anatomy_input = ops.Shared(anatomy_input, sum(2 + (fin != fout) for fin, fout in SPADE_BLOCKS))
    for i, (fin, fout) in enumerate(SPADE_BLOCKS):
        if i > 0:
            l = ops.upsample2(l)
        l = spade_block(m, 'b%d' % i, anatomy_input, l, fin, fout)
    return l
