"""16-bit tensors in HBM (reduced-precision STORAGE, a build-defined extension for BASELINE configs #3 / #5): every `_t` entry
point against the fp32-storage path of the same precision mode on inputs that are already representable in the 16-bit type.
The 16-bit MFMA operands are then identical, the accumulation order is the same kernel's, so a 16-bit OUTPUT must be bit for bit
the rounding of the fp32-storage output (and an fp32 output bit for bit the same)."""
import numpy as np
import pytest
import torch

from multimodal_segmentation_amd import _native as N, ops as P

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


@pytest.fixture(params=['bf16', 'fp16'])
def mode(request):
    prev = P.set_conv_precision(request.param)
    yield (torch.bfloat16 if request.param == 'bf16' else torch.float16)
    P.set_conv_precision(prev)


@pytest.mark.parametrize('B,H,C1,C2,Cout,ups,act', [(2, 32, 64, 0, 64, 0, 1), (2, 16, 128, 0, 64, 1, 0), (1, 32, 64, 64, 128, 0, 1),
                                                    (2, 24, 64, 0, 8, 0, 0)])
def test_conv_fwd_16bit_io(B, H, C1, C2, Cout, ups, act, mode):
    k = 1 if Cout == 8 else 3
    H1 = H // 2 if ups else H
    x1 = rnd(B, H1, H1, C1, seed=1).to(mode).to(DEV)
    x2 = rnd(B, H, H, C2, seed=2).to(mode).to(DEV) if C2 else None
    Cin = C1 + C2
    w = (rnd(k, k, Cin, Cout, seed=3) * 0.05).to(DEV)
    b = rnd(Cout, seed=4).to(DEV)
    wp = torch.empty(w.numel(), device=DEV)
    N.call('mmseg_conv2d_wprep', w, wp, k, k, Cin, Cout, 0)
    p = k // 2
    # fp32-storage path of the same precision mode
    y32 = torch.empty(B, H, H, Cout, device=DEV)
    N.call('mmseg_conv2d_fwd', x1.float(), x2.float() if C2 else None, w, wp, b, y32, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, p, p, ups,
           0, act, 0.0, 0)
    for out16 in (True, False):
        y = torch.empty(B, H, H, Cout, device=DEV, dtype=mode if out16 else torch.float32)
        io = 1 | (2 if C2 else 0) | (4 if out16 else 0)
        N.call('mmseg_conv2d_fwd_t', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, p, p, ups, 0, act, 0.0, 0, io)
        if Cout == 8 and k == 1:
            # the fp32-storage call of this 1x1 head runs on the bandwidth kernel of csrc/smallconv.hpp (round 3), the 16-bit-input call on
            # the MFMA kernel: same rounded operands and fp32 accumulation, another summation order
            ref = y32.to(mode).float() if out16 else y32
            assert (y.float() - ref).abs().max() <= (2.0 ** -7 if out16 else 1e-5) * float(ref.abs().max()), 'out16=%s' % out16
            continue
        assert torch.equal(y, y32.to(mode) if out16 else y32), 'out16=%s' % out16
    # mixed: fp32 inputs, 16-bit output
    y = torch.empty(B, H, H, Cout, device=DEV, dtype=mode)
    N.call('mmseg_conv2d_fwd_t', x1.float(), x2.float() if C2 else None, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, p, p, ups,
           0, act, 0.0, 0, 4)
    if Cout == 8 and k == 1:
        assert (y.float() - y32).abs().max() <= 2.0 ** -7 * float(y32.abs().max())
    else:
        assert torch.equal(y, y32.to(mode))


def test_generic_kernel_writes_16bit_and_refuses_16bit_inputs(mode):
    B, H, Cin, Cout = 2, 20, 3, 64         # (3 input channels: the generic implicit-GEMM kernel, not a small-channel bandwidth kernel)
    x = rnd(B, H, H, Cin, seed=5).to(DEV)
    w = (rnd(3, 3, Cin, Cout, seed=6) * 0.2).to(DEV)
    y32 = torch.empty(B, H, H, Cout, device=DEV)
    N.call('mmseg_conv2d_fwd', x, None, w, None, None, y32, None, B, H, H, Cin, 0, H, H, Cout, 3, 3, 1, 1, 1, 0, 0, 1, 0.0, 0)
    y = torch.empty(B, H, H, Cout, device=DEV, dtype=mode)
    N.call('mmseg_conv2d_fwd_t', x, None, w, None, None, y, None, B, H, H, Cin, 0, H, H, Cout, 3, 3, 1, 1, 1, 0, 0, 1, 0.0, 0, 4)
    assert torch.equal(y, y32.to(mode))
    with pytest.raises(N.NativeLibraryError):      # a 16-bit INPUT needs the MFMA fast path
        N.call('mmseg_conv2d_fwd_t', x.to(mode), None, w, None, None, y, None, B, H, H, Cin, 0, H, H, Cout, 3, 3, 1, 1, 1, 0, 0, 1, 0.0, 0, 5)


def test_16bit_io_needs_a_reduced_precision_mode():
    assert P.set_conv_precision('fp32') in ('fp32', 'bf16', 'fp16')
    x = rnd(1, 16, 16, 64).to(DEV)
    w = (rnd(3, 3, 64, 64) * 0.05).to(DEV)
    wp = torch.empty(w.numel(), device=DEV)
    N.call('mmseg_conv2d_wprep', w, wp, 3, 3, 64, 64, 0)
    y = torch.empty(1, 16, 16, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(N.NativeLibraryError):
        N.call('mmseg_conv2d_fwd_t', x, None, w, wp, None, y, None, 1, 16, 16, 64, 0, 16, 16, 64, 3, 3, 1, 1, 1, 0, 0, 0, 0.0, 0, 4)


@pytest.mark.parametrize('H,C1,C2,Cout,ups', [(32, 64, 0, 64, 0), (16, 128, 128, 128, 0), (32, 64, 0, 128, 0), (16, 128, 0, 64, 1)])
def test_wgrad_16bit_operands(H, C1, C2, Cout, ups, mode):
    B = 2
    H1 = H // 2 if ups else H
    x1 = rnd(B, H1, H1, C1, seed=11).to(mode).to(DEV)
    x2 = rnd(B, H, H, C2, seed=12).to(mode).to(DEV) if C2 else None
    dy = rnd(B, H, H, Cout, seed=13).to(mode).to(DEV)
    Cin = C1 + C2
    need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, Cin, Cout, 3, 3)
    ws = torch.empty(max(need, 1), device=DEV)
    ref = torch.zeros(3, 3, Cin, Cout, device=DEV)
    N.call('mmseg_conv2d_wgrad', x1.float(), x2.float() if C2 else None, dy.float(), ref.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout,
           3, 3, 1, 1, 1, ups, 0)
    for io in (5, 1, 4):
        got = torch.zeros_like(ref)
        a1 = x1 if io & 1 else x1.float()
        a2 = (x2 if io & 1 else x2.float()) if C2 else None
        g = dy if io & 4 else dy.float()
        N.call('mmseg_conv2d_wgrad_t', a1, a2, g, got.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 0, io)
        assert torch.equal(got, ref), 'io=%d: max diff %g' % (io, float((got - ref).abs().max()))


@pytest.mark.parametrize('hx,hy', [(0, 1), (1, 1), (1, 0)])
def test_batchnorm_maxpool_upsample_typed_io(hx, hy, mode):
    """the typed-I/O kernels against the fp32 kernels on representable inputs: statistics and the fp32 results bit-identical up to
    the final rounding of 16-bit outputs"""
    B, H, C = 2, 16, 128
    M = B * H * H
    tx = mode if hx else torch.float32
    ty = mode if hy else torch.float32
    x = (rnd(B, H, H, C, seed=21) * 1.5 + 0.3).to(mode).to(DEV)         # representable in the 16-bit type
    gamma, beta = (rnd(C, seed=22) * 0.2 + 1).to(DEV), (rnd(C, seed=23) * 0.1).to(DEV)
    hcode = lambda f: (1 if mode == torch.bfloat16 else 2) if f else 0
    wsn = torch.empty(N.call('mmseg_norm_workspace_floats', C), device=DEV)
    st_ref, st = torch.empty(4, C, device=DEV), torch.empty(4, C, device=DEV)
    mm, mv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mm2, mv2 = mm.clone(), mv.clone()
    xf = x.float()
    N.call('mmseg_bn_stats', xf, gamma, beta, st_ref[0], st_ref[1], st_ref[2], st_ref[3], mm, mv, wsn, M, C, 1e-3, 0.99)
    xin = x.to(tx)
    N.call('mmseg_bn_stats_t', xin, gamma, beta, st[0], st[1], st[2], st[3], mm2, mv2, wsn, M, C, 1e-3, 0.99, hcode(hx))
    assert torch.equal(st, st_ref) and torch.equal(mm, mm2) and torch.equal(mv, mv2)
    y_ref = torch.empty_like(xf)
    N.call('mmseg_bn_apply', xf, st[2], st[3], y_ref, M, C, 1)
    y = torch.empty(B, H, H, C, device=DEV, dtype=ty)
    N.call('mmseg_bn_apply_t', xin, st[2], st[3], y, M, C, 1, hcode(hx), hcode(hy))
    assert torch.equal(y, y_ref.to(ty))
    # backward on a representable upstream gradient and the (possibly rounded) y
    dy = rnd(B, H, H, C, seed=24).to(mode).to(DEV)
    yq = y.float()                                                       # what the backward pass sees as y
    dx_ref, dg_ref, db_ref = torch.empty_like(xf), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    coef = torch.empty(3 * C, device=DEV)
    N.call('mmseg_bn_bwd', dy.float(), yq, xf, gamma, st[0], st[1], dx_ref, dg_ref, db_ref, coef, wsn, M, C, 1, 1)
    dx, dg, db = torch.empty(B, H, H, C, device=DEV, dtype=tx), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    N.call('mmseg_bn_bwd_t', dy.to(ty), y, xin, gamma, st[0], st[1], dx, dg, db, coef, wsn, M, C, 1, 1, hcode(hx), hcode(hy))
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    assert torch.equal(dx, dx_ref.to(tx))
    if hx == hy and hx:
        # pooling and the up-sampling gradient on 16-bit tensors
        p_ref = torch.empty(B, H // 2, H // 2, C, device=DEV)
        N.call('mmseg_maxpool2_fwd', xf, p_ref, B, H, H, C)
        pq = torch.empty(B, H // 2, H // 2, C, device=DEV, dtype=mode)
        N.call('mmseg_maxpool2_fwd_t', x, pq, B, H, H, C, hcode(1))
        assert torch.equal(pq, p_ref.to(mode))
        gp = rnd(B, H // 2, H // 2, C, seed=25).to(mode).to(DEV)
        dxp_ref = torch.empty_like(xf)
        N.call('mmseg_maxpool2_bwd', xf, p_ref, gp.float(), dxp_ref, B, H, H, C)
        dxp = torch.empty_like(x)
        N.call('mmseg_maxpool2_bwd_t', x, pq, gp, dxp, B, H, H, C, hcode(1))
        assert torch.equal(dxp, dxp_ref.to(mode))
        u_ref = torch.empty(B, H // 2, H // 2, C, device=DEV)
        N.call('mmseg_upsample2_bwd', dy.float(), u_ref, B, H // 2, H // 2, C)
        u = torch.empty(B, H // 2, H // 2, C, device=DEV, dtype=mode)
        N.call('mmseg_upsample2_bwd_t', dy, u, B, H // 2, H // 2, C, hcode(1))
        assert torch.equal(u, u_ref.to(mode))


@pytest.mark.parametrize('act', [1, 2, 3])
@pytest.mark.parametrize('C,bias', [(128, True), (64, True), (24, False)])
def test_activation_gradient_16bit(C, bias, act, mode):
    """mmseg_act_bwd_bias_t with 16-bit dy / y / dx (the gradient of an activated convolution whose output is stored in 16 bits --
    the SPADE units' 128-channel hidden tensor) against act_bwd + colsum on the widened tensors: dx is bit for bit the rounding of the
    fp32 result, the bias gradient the column sums of the stored (rounded) values."""
    M = 3 * 40 * 40
    alpha = 0.2
    y = rnd(M, C, seed=21).to(mode).to(DEV)
    if act == 3:
        y = torch.tanh(y.float()).to(mode)
    dy = rnd(M, C, seed=22).to(mode).to(DEV)
    ref = torch.empty(M, C, device=DEV)
    N.call('mmseg_act_bwd', dy.float(), y.float(), ref, M * C, act, alpha)
    dx = torch.full((M, C), float('nan'), device=DEV, dtype=mode)
    h = 1 if mode == torch.bfloat16 else 2
    if bias:
        bg = torch.full((C,), 0.5, device=DEV)
        ws = torch.empty(N.call('mmseg_colsum_workspace_floats', M, C), device=DEV)
        N.call('mmseg_act_bwd_bias_t', dy, y, dx, bg, ws, M, C, act, alpha, 1, h)
        want = 0.5 + ref.to(mode).double().sum(0)      # the bias gradient sums the STORED values (what the weight / data gradients read)
        assert (bg.double() - want).abs().max() <= 2e-4 * max(1.0, float(want.abs().max()))
    else:
        N.call('mmseg_act_bwd_bias_t', dy, y, dx, None, None, M, C, act, alpha, 0, h)
    assert torch.equal(dx, ref.to(mode))


def test_activated_conv_with_16bit_output_has_a_correct_backward(mode):
    """ops.conv2d(act='relu', out_dtype=half) -- `spade_hidden`: 8 -> 128, the only activated convolution stored in 16 bits -- forward
    and backward through the autograd node against the fp32-storage node (round-2 defect: the activation gradient was written in
    16 bits into an fp32 buffer and read back as fp32 by the weight- and data-gradient launches)."""
    B, H, Cin, Cout = 2, 32, 8, 128
    x = rnd(B, H, H, Cin, seed=31).to(mode).float().to(DEV)
    w = (rnd(3, 3, Cin, Cout, seed=32) * 0.1).to(DEV)
    b = (rnd(Cout, seed=33) * 0.1).to(DEV)
    dy = rnd(B, H, H, Cout, seed=34).to(mode).to(DEV)
    out = {}
    for dt in (torch.float32, mode):
        xg = x.clone().requires_grad_(True)
        wg, bg = torch.zeros_like(w), torch.zeros_like(b)
        anchor = torch.zeros(1, device=DEV, requires_grad=True)
        y = P.conv2d(xg, w, b, 1, 'same', 'relu', 0.0, wgrad=wg, bgrad=bg, anchor=anchor, out_dtype=dt)
        assert y.dtype == dt
        y.backward(dy.to(dt))
        out[dt] = (y.float(), xg.grad.float(), wg, bg)
    for a, b_, name, tol in zip(out[torch.float32], out[mode], ('y', 'dx', 'dw', 'db'), (1e-2, 2e-2, 2e-2, 2e-2)):
        err = float((a - b_).norm() / a.norm())
        assert err <= tol, '%s: rel L2 %.3e' % (name, err)


def test_spade_gamma_beta_tensor_in_16_bits(mode):
    """the fused gamma / beta tensor of a SPADE unit and its gradient stored in 16 bits (conf.act_storage = 'half'): the fused convolution
    writes exactly the rounding of its fp32-storage output; InstanceNorm + modulation read a 16-bit gb and write a 16-bit dgb that is
    bit for bit the rounding of the fp32 one (same arithmetic, only the loads and stores differ); the typed column sums equal the
    fp32 ones on the widened tensor"""
    B, H, Cin, f = 2, 24, 128, 32
    a = rnd(B, H, H, Cin, seed=41).to(mode).to(DEV)
    wg = (rnd(3, 3, Cin, f, seed=42) * 0.03).to(DEV)
    wb = (rnd(3, 3, Cin, f, seed=43) * 0.03).to(DEV)
    bg, bb = (rnd(f, seed=44) * 0.1).to(DEV), (rnd(f, seed=45) * 0.1).to(DEV)
    with torch.no_grad():
        gb32 = P.conv2d_pair(a, wg, bg, wb, bb)
        gb16 = P.conv2d_pair(a, wg, bg, wb, bb, out_dtype=mode)
    assert gb16.dtype == mode and torch.equal(gb16, gb32.to(mode))
    x = (rnd(B, H, H, f, seed=46) * 1.5 + 0.2).to(DEV)
    dy = rnd(B, H, H, f, seed=47).to(DEV)
    out = {}
    dy = dy.to(mode).float()                            # a cotangent that is representable in 16 bits
    for gb in (gb16.float(), gb16):                     # the same (representable) values in fp32 and in 16-bit storage
        xg = x.clone().requires_grad_(True)
        gg = gb.clone().requires_grad_(True)
        y = P.instnorm_spade_gb(xg, gg, 0.2)
        y.backward(dy)
        out[gb.dtype] = (y.detach(), xg.grad, gg.grad)
    y32, dx32, dgb32 = out[torch.float32]
    y16, dx16, dgb16 = out[mode]
    assert torch.equal(y16, y32) and torch.equal(dx16, dx32)
    assert dgb16.dtype == mode and torch.equal(dgb16, dgb32.to(mode))
    # ... and with the OUTPUT (and its incoming gradient) stored in 16 bits as well: y is the rounding of the fp32 y, the gradients are
    # those of the fp32-storage call (the cotangent is representable)
    xg = x.clone().requires_grad_(True)
    gg = gb16.clone().requires_grad_(True)
    yh = P.instnorm_spade_gb(xg, gg, 0.2, out_dtype=mode)
    assert yh.dtype == mode and torch.equal(yh, y32.to(mode))
    yh.backward(dy.to(mode))
    assert torch.equal(xg.grad, dx32) and torch.equal(gg.grad, dgb16)
    M, C = B * H * H, 2 * f
    ws = torch.empty(N.call('mmseg_colsum_workspace_floats', M, C), device=DEV)
    o16, o32 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    N.call('mmseg_colsum_t', dgb16, o16, ws, M, C, 0, 1 if mode == torch.bfloat16 else 2)
    N.call('mmseg_colsum', dgb16.float(), o32, ws, M, C, 1.0, 0)
    assert (o16 - o32).abs().max() <= 1e-5 * max(1.0, float(o32.abs().max()))


@pytest.mark.parametrize('dt,decoder', [('bf16', 'film'), ('fp16', 'film'), ('bf16', 'spade')])
def test_generator_step_with_16bit_trunk_storage_close_to_fp32_storage(dt, decoder):
    """conf.act_storage = 'half' at the model level: a teacher-forced DAFNet generator step with the trunk's activations and gradients
    stored in 16 bits against the same step with fp32 storage (same 16-bit MFMA compute mode, same weights and draws): every loss
    term within 3e-2 relative (the only difference is the rounding of the stored tensors), and the 16-bit tensors really exist."""
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    B, H = 2, 64
    d = Hh.make_step_data(B, H, H)
    B1 = np.ones((B, 1), np.float32)
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [B1] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [B1] * 4 + \
         [np.zeros(B, np.float32)] * 2 + [d['z1'], d['z2']]
    res, grads, teacher, ref_w = {}, {}, None, None
    try:
        for storage in ('fp32', 'half'):
            conf = Hh.make_conf(dafnet_config_chaos, H, compute_dtype=dt, act_storage=storage, decoder_type=decoder)
            model = DAFNet(conf)
            model.build()
            assert (P.act16_dtype() is not None) == (storage == 'half')
            ms = model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]
            if ref_w is None:
                ref_w = [m.get_weights() for m in ms]
            else:
                for m, w in zip(ms, ref_w):
                    m.set_weights(w)
            with Hh.teacher_forcing(model, teacher):
                h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
            if teacher is None:
                teacher = [model.last_factors['s1'].detach().clone(), model.last_factors['s2'].detach().clone()]
            res[storage] = {k: h.history[k][0] for k in h.history.keys()}
            grads[storage] = Hh.product_grads(model)
            if storage == 'half':
                # the segmentor's first block hands a 16-bit tensor to its second convolution
                x = torch.zeros(B, H, H, 8, device='cuda')
                with torch.no_grad():
                    l = nn.conv_bn(model.Segmentor, 'c0', 'c0_bn', x, False, relu=True)
                assert l.dtype == P.act16_dtype()
                if decoder == 'spade':
                    # the 128-channel hidden tensor of every SPADE unit (conv 8 -> 128 + ReLU) is stored in 16 bits
                    from multimodal_segmentation_amd.layers import spade as SP
                    with torch.no_grad():
                        a = SP.spade_hidden(model.Decoder, 'b5_s0', x)
                    assert a.dtype == P.act16_dtype() and a.shape[-1] == 128
    finally:
        P.set_activation_storage(False)
        P.set_conv_precision('fp32')
    for k, v in res['fp32'].items():
        tol = 6e-2 if k in ('loss', 'D_Mask_loss', 'D_Image1_loss', 'D_Image2_loss') else 3e-2
        assert np.isfinite(res['half'][k]) and abs(res['half'][k] - v) <= tol * max(1.0, abs(v)), (k, v, res['half'][k])
    assert any(abs(res['half'][k] - v) > 1e-7 for k, v in res['fp32'].items())
    # the BACKWARD pass (the losses above are forward quantities): weight gradients of the 16-bit-storage step against the fp32-storage
    # step, relative L2 per tensor -- for the components DOWNSTREAM of the teacher-forced anatomies (decoder, segmentor, modality
    # encoder, fuser).  The anatomy encoders' gradients are left out: measured against the fp64 oracle they carry 40 % (bf16) / 18 %
    # (fp16) noise at this size in EVERY 16-bit implementation, the fp32-storage oracle included
    # (profiles/r03_gradient_errors_*_bf16_cuda.txt), so they cannot tell a defect from rounding.  Downstream the noise is 4 % (FiLM) /
    # 24 % (SPADE, 30 convolutions with InstanceNorm); a gradient tensor written with the wrong element type (round 2's activated
    # 16-bit convolution `spade_hidden`) is off by O(1) in every SPADE unit.
    worst = []
    for k, g32 in grads['fp32'].items():
        if not k.startswith(('DEC/', 'SEG/', 'EM/', 'FUS/')):
            continue
        g16 = grads['half'][k]
        assert np.isfinite(g16).all(), k
        nrm = np.linalg.norm(g32)
        if nrm < 1e-9:
            continue
        worst.append((float(np.linalg.norm(g16 - g32) / nrm), k))
    worst.sort(reverse=True)
    print('worst downstream gradient rel-L2 (16-bit vs fp32 storage):', worst[:6])
    assert worst[0][0] <= (0.5 if decoder == 'spade' else 0.15), worst[:6]


@pytest.mark.parametrize('B,H,W,C1,C2,Cout,k,ups,act', [
    (2, 32, 32, 64, 0, 64, 3, 0, 1),        # 64-wide tile, 2-D pixel tiles
    (1, 32, 48, 128, 0, 128, 3, 0, 0),      # 128-wide tile
    (1, 16, 16, 128, 0, 256, 3, 0, 1),      # 256-wide tile, one pixel tile
    (3, 12, 10, 64, 0, 192, 3, 0, 2),       # raster rows, M = 360 (partial second tile), Cout between tile widths
    (2, 16, 16, 128, 0, 64, 3, 1, 0),       # nearest x2 up-sampling folded into the gather
    (1, 32, 32, 64, 64, 128, 3, 0, 1),      # two inputs (skip concatenation): K tiles walk x1, then x2
    (2, 16, 16, 192, 0, 320, 1, 0, 0),      # 1x1, Cout not a multiple of the tile
    (1, 20, 20, 64, 0, 72, 5, 0, 0),        # 5x5 taps, Cout = 72
    # 3x3 'same' with W % 32 == 0 and H % 8 == 0: the patch-resident kernel (conv16h_kernel, family 17)
    (2, 8, 32, 64, 0, 128, 3, 0, 2),        # one tile per image: every patch border lies outside the image
    (1, 24, 64, 128, 0, 256, 3, 0, 1),      # 3 x 2 tiles, interior borders, 256-wide N tile (2 weight stages)
    (3, 16, 32, 64, 0, 192, 3, 0, 0),       # Cout between the tile widths, batch 3
    (2, 16, 32, 128, 0, 64, 3, 1, 1),       # up-sampled x1 (stored 8 x 16)
    (1, 16, 64, 64, 128, 64, 3, 0, 0),      # concatenation: chunks walk x1 (1 chunk) then x2 (2 chunks)
    (1, 32, 32, 192, 0, 320, 3, 0, 0),      # three chunks, two N tiles of 256
    (2, 32, 64, 64, 0, 64, 3, 0, 1),        # 64-channel form: 16 rows x 32 columns per block, one chunk
    (1, 16, 32, 64, 128, 56, 3, 0, 2),      # ... three chunks (single patch buffer reloaded twice), Cout = 56
    (2, 32, 32, 128, 0, 64, 3, 1, 0),       # ... up-sampled input, two chunks
])
def test_large_tile_16bit_kernel_equals_the_register_staged_kernel(B, H, W, C1, C2, Cout, k, ups, act, mode):
    """conv16.hpp (256-pixel tiles, buffer_load ... lds, ring of LDS stages) forced onto small problems (mmseg_conv16_mode 2) against
    conv_fast_kernel (mode 0): the same 16-bit operands on the same MFMA, sums associated differently -> equal to fp32 rounding;
    and against the fp64 oracle on the rounded operands."""
    from oracle import ops as O
    H1, W1 = (H // 2, W // 2) if ups else (H, W)
    x1 = rnd(B, H1, W1, C1, seed=1).to(mode).to(DEV)
    x2 = rnd(B, H, W, C2, seed=2).to(mode).to(DEV) if C2 else None
    Cin = C1 + C2
    w = (rnd(k, k, Cin, Cout, seed=3) * (2.0 / (k * k * Cin)) ** 0.5).to(DEV)
    b = rnd(Cout, seed=4).to(DEV)
    wp = torch.empty(w.numel(), device=DEV)
    N.call('mmseg_conv2d_wprep', w, wp, k, k, Cin, Cout, 0)
    p = k // 2
    io_in = 1 | (2 if C2 else 0)
    prev = N.call('mmseg_conv16_mode', 0)
    try:
        outs = {}
        for m16 in (0, 2):
            N.call('mmseg_conv16_mode', m16)
            for out16 in (False, True):
                y = torch.full((B, H, W, Cout), float('nan'), device=DEV, dtype=mode if out16 else torch.float32)
                N.call('mmseg_conv2d_fwd_t', x1, x2, w, wp, b, y, None, B, H, W, C1, C2, H, W, Cout, k, k, 1, p, p, ups, 0, act, 0.2, 0,
                       io_in | (4 if out16 else 0))
                outs[(m16, out16)] = y
                fam = N.call('mmseg_conv2d_last_kernel') // 1000000
                hres = k == 3 and W % 32 == 0 and H % (16 if Cout <= 64 else 8) == 0
                assert fam == ((17 if hres else 16) if m16 == 2 else 1), 'launch went to kernel family %d' % fam
        # same operands, same MFMA; the K tiles are 64 channels deep here and 32 or 64 in the other kernel (its choice depends on the
        # tile): the fp32 sums differ by their association only
        scale = float(outs[(0, False)].abs().max())
        assert not torch.isnan(outs[(2, False)]).any() and not torch.isnan(outs[(2, True)].float()).any()
        assert float((outs[(0, False)] - outs[(2, False)]).abs().max()) <= 1e-5 * scale
        assert float((outs[(0, True)].float() - outs[(2, True)].float()).abs().max()) <= 2.0 ** -7 * scale
        assert torch.equal(outs[(2, True)], outs[(2, False)].to(mode)), 'a 16-bit output is the rounding of the fp32 output'
    finally:
        N.call('mmseg_conv16_mode', prev)
    # oracle on the operands as the MFMA sees them (16-bit inputs are exact; the weight image is rounded to the 16-bit type)
    a = x1.float().cpu().double()
    if ups:
        a = O.upsample2(a)
    if C2:
        a = torch.cat([a, x2.float().cpu().double()], -1)
    ref = O.conv2d(a, w.to(mode).float().cpu().double(), b.cpu().double())
    ref = torch.relu(ref) if act == 1 else (O.leaky_relu(ref, 0.2) if act == 2 else ref)
    err = float((outs[(2, False)].cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 4e-4, err


def test_large_tile_16bit_kernel_split_output_and_scale(mode):
    """the two-output epilogue (data gradient of a convolution over concatenated inputs: channels [0, n1) -> y, the rest -> y2) and the
    per-channel scale of the BatchNorm-folded inference convolution, large-tile kernel vs register-staged kernel"""
    B, H, Cin, C1o, C2o = 2, 16, 128, 64, 64
    x = rnd(B, H, H, Cin, seed=1).to(mode).to(DEV)
    w = (rnd(3, 3, Cin, C1o + C2o, seed=3) * 0.03).to(DEV)
    wp = torch.empty(w.numel(), device=DEV)
    N.call('mmseg_conv2d_wprep', w, wp, 3, 3, Cin, C1o + C2o, 0)
    scale, shift = (rnd(C1o + C2o, seed=5) * 0.1 + 1).to(DEV), rnd(C1o + C2o, seed=6).to(DEV)
    prev = N.call('mmseg_conv16_mode', 0)
    try:
        res = {}
        for m16 in (0, 2):
            N.call('mmseg_conv16_mode', m16)
            y1 = torch.empty(B, H, H, C1o, device=DEV, dtype=mode)
            y2 = torch.empty(B, H, H, C2o, device=DEV, dtype=mode)
            N.call('mmseg_conv2d_fwd_t', x, None, w, wp, None, y1, y2, B, H, H, Cin, 0, H, H, C1o + C2o, 3, 3, 1, 1, 1, 0, 0, 0, 0.0, C1o, 1 | 4)
            ys = torch.empty(B, H, H, C1o + C2o, device=DEV, dtype=mode)
            N.call('mmseg_conv2d_fwd_scaled_t', x, None, w, wp, shift, scale, ys, B, H, H, Cin, 0, H, H, C1o + C2o, 3, 3, 1, 1, 1, 0, 1, 0.0, 1 | 4)
            res[m16] = (y1, y2, ys)
        for a, c in zip(res[0], res[2]):
            assert float((a.float() - c.float()).abs().max()) <= 2.0 ** -7 * float(a.float().abs().max())
    finally:
        N.call('mmseg_conv16_mode', prev)


@pytest.mark.parametrize('B,H,W,C1,C2,Cout,ups,act', [
    (2, 32, 64, 64, 0, 64, 0, 1),           # 64-channel form (16 rows per block, single patch buffer, two 32-channel chunks)
    (1, 16, 32, 32, 0, 128, 0, 0),          # 128 channels, 16 rows
    (1, 24, 64, 64, 0, 256, 0, 1),          # 256 channels, 8 rows, 3 x 2 tiles
    (2, 16, 32, 64, 0, 64, 1, 1),           # up-sampled input
    (1, 16, 64, 32, 64, 64, 0, 0),          # concatenation: one chunk of x1, two of x2
    (1, 32, 32, 96, 0, 320, 0, 0),          # three chunks, Cout beyond one N tile
    (1, 16, 32, 64, 32, 56, 0, 2),          # Cout = 56, LeakyReLU
    (3, 8, 32, 32, 0, 128, 0, 0),           # 8-row image: the 8-row form of the 128-channel tile
])
def test_fp32_patch_resident_kernel(B, H, W, C1, C2, Cout, ups, act):
    """conv16h_kernel<..., PREC 0, ...> (round 4: the patch-resident large-tile kernel on v_mfma_f32_32x32x2_f32, rows of 32 fp32
    channels) forced onto small problems against conv_fast_kernel and against the fp64 oracle: fp32 products, fp32 accumulation,
    another association of the sums"""
    from oracle import ops as O
    prevp = P.set_conv_precision('fp32')
    prev = N.call('mmseg_conv16_mode', 0)
    try:
        H1, W1 = (H // 2, W // 2) if ups else (H, W)
        x1 = rnd(B, H1, W1, C1, seed=1).to(DEV)
        x2 = rnd(B, H, W, C2, seed=2).to(DEV) if C2 else None
        Cin = C1 + C2
        w = (rnd(3, 3, Cin, Cout, seed=3) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
        b = rnd(Cout, seed=4).to(DEV)
        wp = torch.empty(w.numel(), device=DEV)
        N.call('mmseg_conv2d_wprep', w, wp, 3, 3, Cin, Cout, 0)
        outs = {}
        for m16 in (0, 2):
            N.call('mmseg_conv16_mode', m16)
            y = torch.full((B, H, W, Cout), float('nan'), device=DEV)
            N.call('mmseg_conv2d_fwd', x1, x2, w, wp, b, y, None, B, H, W, C1, C2, H, W, Cout, 3, 3, 1, 1, 1, ups, 0, act, 0.2, 0)
            outs[m16] = y
            assert N.call('mmseg_conv2d_last_kernel') // 1000000 == (17 if m16 == 2 else 1)
        a = x1.cpu().double()
        if ups:
            a = O.upsample2(a)
        if C2:
            a = torch.cat([a, x2.cpu().double()], -1)
        ref = O.conv2d(a, w.cpu().double(), b.cpu().double())
        ref = torch.relu(ref) if act == 1 else (O.leaky_relu(ref, 0.2) if act == 2 else ref)
        scale = float(ref.abs().max())
        assert not torch.isnan(outs[2]).any()
        assert float((outs[2].cpu().double() - ref).abs().max()) <= 5e-6 * scale
        assert float((outs[2] - outs[0]).abs().max()) <= 5e-6 * scale
    finally:
        N.call('mmseg_conv16_mode', prev)
        P.set_conv_precision(prevp)


@pytest.mark.parametrize('B,H,W,C1,C2,Cout,ups,acc', [
    (2, 8, 32, 32, 0, 128, 0, 0),           # <1, 4>: one input-channel plane, four output-channel planes
    (1, 16, 64, 64, 0, 64, 0, 1),           # <2, 2>, accumulating into dW
    (2, 6, 32, 128, 0, 32, 0, 0),           # <4, 1>
    (1, 8, 64, 64, 64, 64, 0, 0),           # concatenated inputs: planes walk x1, then x2
    (2, 8, 32, 64, 0, 128, 1, 1),           # up-sampled x1 (stored 4 x 16)
    (3, 4, 32, 96, 0, 256, 0, 0),           # three input planes, two output tiles, batch 3
])
def test_fp32_patch_resident_weight_gradient(B, H, W, C1, C2, Cout, ups, acc):
    """wgrad32h_kernel (round 4: the 3x3 weight gradient with the activation patch and the dy tile resident in LDS, operands read as
    they lie -- no transposition on the fp32 MFMA) forced onto small problems (mmseg_conv16_mode 2) against conv_wgrad_tr_kernel
    (mode 0) and against the fp64 oracle's autograd"""
    from oracle import ops as O
    prevp = P.set_conv_precision('fp32')
    prev = N.call('mmseg_conv16_mode', 0)
    try:
        H1, W1 = (H // 2, W // 2) if ups else (H, W)
        x1 = rnd(B, H1, W1, C1, seed=1).to(DEV)
        x2 = rnd(B, H, W, C2, seed=2).to(DEV) if C2 else None
        dy = rnd(B, H, W, Cout, seed=3).to(DEV)
        Cin = C1 + C2
        base = (rnd(3, 3, Cin, Cout, seed=4) * 0.1).to(DEV)
        need = N.call('mmseg_conv2d_wgrad_workspace', B, H, W, Cin, Cout, 3, 3)
        ws = torch.full((max(need, 1),), float('nan'), device=DEV)
        outs = {}
        for m16 in (0, 2):
            N.call('mmseg_conv16_mode', m16)
            dw = base.clone() if acc else torch.full_like(base, float('nan'))
            N.call('mmseg_conv2d_wgrad', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, W, C1, C2, H, W, Cout, 3, 3, 1, 1, 1, ups, acc)
            outs[m16] = dw
            fam = N.call('mmseg_conv2d_last_kernel') // 1000000
            assert (fam == 18) == (m16 == 2), 'launch went to kernel family %d' % fam
        a = x1.cpu().double()
        if ups:
            a = O.upsample2(a)
        if C2:
            a = torch.cat([a, x2.cpu().double()], -1)
        wref = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
        O.conv2d(a, wref, None).backward(dy.cpu().double())
        ref = wref.grad + (base.cpu().double() if acc else 0.0)
        scale = float(ref.abs().max())
        assert not torch.isnan(outs[2]).any()
        assert float((outs[2].cpu().double() - ref).abs().max()) <= 2e-5 * scale
        assert float((outs[2] - outs[0]).abs().max()) <= 2e-5 * scale
        # bitwise reproducible: fixed tile order per block, fixed-order slab reduction
        dw2 = base.clone() if acc else torch.full_like(base, float('nan'))
        N.call('mmseg_conv2d_wgrad', x1, x2, dy, dw2.view(-1), ws, ws.numel(), B, H, W, C1, C2, H, W, Cout, 3, 3, 1, 1, 1, ups, acc)
        assert torch.equal(dw2, outs[2])
    finally:
        N.call('mmseg_conv16_mode', prev)
        P.set_conv_precision(prevp)


@pytest.mark.parametrize('B,H,W,C1,C2,Cout,ups,acc', [
    (2, 8, 32, 64, 0, 64, 0, 0),            # one tile per image: every patch border outside the image
    (1, 16, 64, 64, 0, 128, 0, 1),          # 2 x 2 tiles, two output planes, accumulating
    (2, 8, 32, 128, 0, 64, 1, 0),           # up-sampled x1, two input planes
    (1, 24, 32, 64, 64, 64, 0, 0),          # concatenated inputs, three tiles down one column strip
    (3, 8, 64, 192, 0, 128, 0, 1),          # three input planes, batch 3
    (2, 16, 32, 128, 0, 32, 0, 0),          # Cout = 32: one half-filled output plane (the fused gamma / beta convolution of a 16-channel SPADE unit)
    (1, 8, 64, 64, 0, 96, 0, 1),            # Cout = 96: a full and a half-filled plane
])
def test_16bit_patch_resident_weight_gradient(B, H, W, C1, C2, Cout, ups, acc, mode):
    """wgrad16h_kernel (round 4: operands through ds_read_b64_tr_b16 from the [pixel][channel] image in LDS) forced onto small problems
    against conv_wgrad_tr_kernel's 16-bit instance (mode 0) and against the fp64 oracle on the 16-bit operands"""
    from oracle import ops as O
    prev = N.call('mmseg_conv16_mode', 0)
    try:
        H1, W1 = (H // 2, W // 2) if ups else (H, W)
        x1 = rnd(B, H1, W1, C1, seed=1).to(mode).to(DEV)
        x2 = rnd(B, H, W, C2, seed=2).to(mode).to(DEV) if C2 else None
        dy = rnd(B, H, W, Cout, seed=3).to(mode).to(DEV)
        Cin = C1 + C2
        base = (rnd(3, 3, Cin, Cout, seed=4) * 0.1).to(DEV)
        need = N.call('mmseg_conv2d_wgrad_workspace', B, H, W, Cin, Cout, 3, 3)
        ws = torch.full((max(need, 1),), float('nan'), device=DEV)
        outs = {}
        for m16 in (0, 2):
            N.call('mmseg_conv16_mode', m16)
            dw = base.clone() if acc else torch.full_like(base, float('nan'))
            N.call('mmseg_conv2d_wgrad_t', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, W, C1, C2, H, W, Cout, 3, 3, 1, 1, 1, ups, acc, 5)
            outs[m16] = dw
            fam = N.call('mmseg_conv2d_last_kernel') // 1000000
            assert (fam == 19) == (m16 == 2), 'launch went to kernel family %d' % fam
        a = x1.float().cpu().double()
        if ups:
            a = O.upsample2(a)
        if C2:
            a = torch.cat([a, x2.float().cpu().double()], -1)
        wref = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
        O.conv2d(a, wref, None).backward(dy.float().cpu().double())
        ref = wref.grad + (base.cpu().double() if acc else 0.0)
        scale = float(ref.abs().max())
        assert not torch.isnan(outs[2]).any()
        assert float((outs[2].cpu().double() - ref).abs().max()) <= 2e-5 * scale       # exact 16-bit products, fp32 sums
        assert float((outs[2] - outs[0]).abs().max()) <= 2e-5 * scale
        dw2 = base.clone() if acc else torch.full_like(base, float('nan'))
        N.call('mmseg_conv2d_wgrad_t', x1, x2, dy, dw2.view(-1), ws, ws.numel(), B, H, W, C1, C2, H, W, Cout, 3, 3, 1, 1, 1, ups, acc, 5)
        assert torch.equal(dw2, outs[2])
    finally:
        N.call('mmseg_conv16_mode', prev)


@pytest.mark.parametrize('B,H,W,Cout,act,x16', [
    (2, 7, 32, 128, 1, False),       # the SPADE unit's shared convolution: fp32 anatomy, ReLU, one 128-wide column group
    (1, 16, 64, 128, 1, True),       # 16-bit input rows (16 bytes per pixel), two segments per image row
    (3, 5, 96, 64, 0, False),        # 64 output channels: half of the column group is masked
    (1, 9, 32, 320, 2, True),        # three column groups, the last one partly filled (waves of different groups interleave)
    (2, 3, 32, 40, 2, False),        # Cout % 8 == 0 only; LeakyReLU
])
def test_conv_of_8_channels_in_one_launch(B, H, W, Cout, act, x16, mode):
    """conv8h_kernel (conv16.hpp): Conv2D(Cout, 3, 'same') of an 8-channel tensor -- a lane's MFMA operand is one tap of one pixel read as it
    lies, the weights stay in registers -- against the fp64 oracle on the operands as the MFMA sees them (inputs and weights rounded to
    the 16-bit type: exact products, fp32 sums), with fp32 and 16-bit output; every image border is an out-of-range buffer offset."""
    from oracle import ops as O
    x = rnd(B, H, W, 8, seed=1).to(mode)
    xd = (x if x16 else x.float()).to(DEV)
    w = (rnd(3, 3, 8, Cout, seed=2) * (2.0 / 72) ** 0.5).to(DEV)
    b = rnd(Cout, seed=3).to(DEV)
    hm = 1 if mode == torch.bfloat16 else 2
    ys = {}
    for out16 in (False, True):
        y = torch.full((B, H, W, Cout), float('nan'), device=DEV, dtype=mode if out16 else torch.float32)
        N.call('mmseg_conv8h_fwd_t', xd, w, b, y, B, H, W, Cout, act, 0.2, hm if x16 else 0, hm if out16 else 0)
        assert N.call('mmseg_conv2d_last_kernel') // 1000000 == 20
        ys[out16] = y
    assert not torch.isnan(ys[False]).any()
    assert torch.equal(ys[True], ys[False].to(mode)), 'a 16-bit output is the rounding of the fp32 output'
    ref = O.conv2d(x.float().double(), w.to(mode).float().cpu().double(), b.cpu().double())
    ref = torch.relu(ref) if act == 1 else (O.leaky_relu(ref, 0.2) if act == 2 else (torch.tanh(ref) if act == 3 else ref))
    err = float((ys[False].cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err
    # without a bias
    y = torch.empty((B, H, W, Cout), device=DEV)
    N.call('mmseg_conv8h_fwd_t', xd, w, None, y, B, H, W, Cout, 0, 0.0, hm if x16 else 0, 0)
    ref0 = O.conv2d(x.float().double(), w.to(mode).float().cpu().double(), None)
    assert float((y.cpu().double() - ref0).abs().max()) <= 2e-5 * float(ref0.abs().max())


def test_conv_of_8_channels_rejects_what_it_does_not_take(mode):
    x = torch.zeros(1, 4, 48, 8, device=DEV); w = torch.zeros(3, 3, 8, 128, device=DEV); y = torch.zeros(1, 4, 48, 128, device=DEV)
    with pytest.raises(N.NativeLibraryError):
        N.call('mmseg_conv8h_fwd_t', x, w, None, y, 1, 4, 48, 128, 0, 0.0, 0, 0)           # W % 32 != 0
    other = 2 if mode == torch.bfloat16 else 1
    with pytest.raises(N.NativeLibraryError):
        N.call('mmseg_conv8h_fwd_t', x, w, None, y, 1, 4, 32, 128, 0, 0.0, other, 0)       # 16-bit type of the other mode
    with pytest.raises(N.NativeLibraryError):
        N.call('mmseg_conv8h_fwd_t', x, w, None, y, 1, 4, 32, 128, 3, 0.0, 0, 0)           # tanh: the node keeps the im2col path for it


def test_node_of_the_8_channel_conv_uses_the_one_launch_kernel_and_matches_the_im2col_path(mode):
    """ops.conv2d on an 8-channel input in a 16-bit mode: conv8h_kernel when W % 32 == 0 (mmseg_conv16_mode != 0), else / before round 4
    im2col rows + a 1x1 product; same operands, same MFMA, fp32 sums associated differently"""
    B, H, W, Cout = 2, 16, 32, 128
    x = rnd(B, H, W, 8, seed=5).to(DEV)
    w = (rnd(3, 3, 8, Cout, seed=6) * 0.2).to(DEV)
    b = rnd(Cout, seed=7).to(DEV)
    prev = N.call('mmseg_conv16_mode', -1)
    try:
        outs = {}
        for m16 in (0, 1):
            N.call('mmseg_conv16_mode', m16)
            with torch.no_grad():
                outs[m16] = P.conv2d(x, w, b, 1, 'same', 'relu', 0.0, out_dtype=mode)
            fam = N.call('mmseg_conv2d_last_kernel') // 1000000
            assert (fam == 20) == (m16 == 1), fam
        d = (outs[0].float() - outs[1].float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(outs[0].float().abs().max())
        assert float((d > 0).float().mean()) < 0.01       # (a 16-bit rounding flips only where the fp32 sums straddle a tie)
    finally:
        N.call('mmseg_conv16_mode', prev)


@pytest.mark.parametrize('B,H,W,act', [(2, 20, 23, 2), (1, 64, 64, 2), (3, 9, 12, 0), (1, 5, 5, 2)])
def test_locnet_first_layer_in_the_16bit_modes(B, H, W, act, mode):
    """locnet5_fwd_kernel (s2conv.hpp): Conv2D(20, 5, 'valid') + LeakyReLU(0.3) over Concatenate([anatomy 1, anatomy 2]) (stn_spline.py:102-107) --
    fp32 tensors, operands rounded to the 16-bit type on the way into v_mfma_f32_16x16x32_*: against the fp64 oracle on the rounded operands
    (exact products, fp32 sums), pixel counts that fill neither the last 16-pixel tile nor the last pass of two tiles"""
    from oracle import ops as O
    x1 = rnd(B, H, W, 8, seed=1).to(DEV)
    x2 = rnd(B, H, W, 8, seed=2).to(DEV)
    w = (rnd(5, 5, 16, 20, seed=3) * (2.0 / 400) ** 0.5).to(DEV)
    b = rnd(20, seed=4).to(DEV)
    Ho, Wo = H - 4, W - 4
    y = torch.full((B, Ho, Wo, 20), float('nan'), device=DEV)
    N.call('mmseg_conv2d_fwd', x1, x2, w, None, b, y, None, B, H, W, 8, 8, Ho, Wo, 20, 5, 5, 1, 0, 0, 0, 0, act, 0.3, 0)
    assert N.call('mmseg_conv2d_last_kernel') // 1000000 == 23
    assert not torch.isnan(y).any()
    xin = torch.cat([x1.to(mode).float().cpu().double(), x2.to(mode).float().cpu().double()], -1)
    ref = O.conv2d(xin, w.to(mode).float().cpu().double(), b.cpu().double(), padding='valid')
    ref = O.leaky_relu(ref, 0.3) if act == 2 else ref
    err = float((y.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err
    # without a bias
    N.call('mmseg_conv2d_fwd', x1, x2, w, None, None, y, None, B, H, W, 8, 8, Ho, Wo, 20, 5, 5, 1, 0, 0, 0, 0, 0, 0.0, 0)
    ref0 = O.conv2d(xin, w.to(mode).float().cpu().double(), None, padding='valid')
    assert float((y.cpu().double() - ref0).abs().max()) <= 2e-5 * float(ref0.abs().max())


@pytest.mark.parametrize('B,H,W,C,Cout,pad,split', [(2, 18, 22, 20, 20, 0, False), (1, 13, 80, 20, 20, 4, False), (2, 16, 19, 20, 16, 4, True)])
def test_locnet_other_layers_round_their_operands_in_the_16bit_modes(B, H, W, C, Cout, pad, split, mode):
    """locnet5_f32_kernel<..., PREC> (s2conv.hpp): the second / third 5 x 5 layer, its data gradient (padding 4) and the first layer's data
    gradient (16 outputs written to two 8-channel tensors) multiply on the fp32 MFMA in every mode; in the 16-bit modes their operands are
    rounded where the generic kernel's 16-bit instance rounded them -- against the fp64 oracle on the rounded operands"""
    from oracle import ops as O
    x = rnd(B, H, W, C, seed=1).to(DEV)
    w = (rnd(5, 5, C, Cout, seed=2) * (2.0 / (25 * C)) ** 0.5).to(DEV)
    Ho, Wo = H + 2 * pad - 4, W + 2 * pad - 4
    y = torch.full((B, Ho, Wo, 8 if split else Cout), float('nan'), device=DEV)
    y2 = torch.full((B, Ho, Wo, Cout - 8), float('nan'), device=DEV) if split else None
    N.call('mmseg_conv2d_fwd', x, None, w, None, None, y, y2, B, H, W, C, 0, Ho, Wo, Cout, 5, 5, 1, pad, pad, 0, 0, 0, 0.0, 8 if split else 0)
    assert N.call('mmseg_conv2d_last_kernel') // 1000000 == 23
    xin = torch.nn.functional.pad(x.to(mode).float().cpu().double(), (0, 0, pad, pad, pad, pad))
    ref = O.conv2d(xin, w.to(mode).float().cpu().double(), None, padding='valid')
    out = torch.cat([y, y2], -1) if split else y
    assert not torch.isnan(out).any()
    assert float((out.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
