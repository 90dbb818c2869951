"""Parity at BASELINE.json's full size (256x256, batch 8 per GPU): direct oracle comparisons where the oracle finishes in
seconds, and size-independent properties of the domain elsewhere (linearity of the convolutions, fast path == generic
path, BatchNorm moments, softmax/rounding invariants, identity warp, Adam closed form, whole-iteration sanity)."""
import math

import numpy as np
import pytest
import torch

from multimodal_segmentation_amd import nn, ops as P
from oracle import ops as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
B, H = 8, 256


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_unet_d_l0_b_conv_against_oracle():
    """The survey's minimum slice: conv3x3 on [8,256,256,64] -> 64 (UNet d_l0.b), forward + both gradients vs the oracle."""
    x, w, b = rnd(B, H, H, 64, seed=1), rnd(3, 3, 64, 64, seed=2) * 0.06, rnd(64, seed=3) * 0.1
    cot = rnd(B, H, H, 64, seed=4)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = O.conv2d(xr, wr, br)
    yr.backward(cot)
    xd = x.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV), b.to(DEV)
    wg, bg = torch.zeros_like(wd), torch.zeros_like(bd)
    y = P.conv2d(xd, wd, bd, wgrad=wg, bgrad=bg, anchor=nn.anchor(xd.device))
    y.backward(cot.to(DEV))
    assert _rel(y.detach().cpu(), yr.detach()) < 1e-4        # oracle here is fp32 oneDNN: both carry ~1e-6 rounding
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-4
    assert _rel(wg.cpu(), wr.grad) < 1e-3                    # 524288-term fp32 sums
    assert _rel(bg.cpu(), br.grad) < 1e-3


@pytest.mark.parametrize('cin,cout,hw', [(64, 64, 256), (128, 128, 128), (512, 256, 64), (1024, 1024, 16)])
def test_conv_linearity_and_fast_vs_generic(cin, cout, hw):
    """conv(a*x1 + x2) == a*conv(x1) + conv(x2) (bias-free), and the buffer-load fast path equals the generic kernel."""
    from multimodal_segmentation_amd import _native as N
    x1, x2 = rnd(B, hw, hw, cin, seed=5).to(DEV), rnd(B, hw, hw, cin, seed=6).to(DEV)
    w = (rnd(3, 3, cin, cout, seed=7) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    y1, y2 = P.conv2d(x1, w), P.conv2d(x2, w)
    ysum = P.conv2d(P.axpby(x1, x2, 0.5, 1.0), w)
    assert _rel(ysum, P.axpby(y1, y2, 0.5, 1.0)) < 2e-5
    yg = torch.empty_like(y1)                                 # wt = None forces the generic kernel
    N.call('mmseg_conv2d_fwd', x1, None, w, None, None, yg, None, B, hw, hw, cin, 0, hw, hw, cout, 3, 3, 1, 1, 1, 0, 0, 0, 0.0, 0)
    assert _rel(y1, yg) < 2e-5


def test_strided_dgrad_parity_classes_equal_fractionally_strided():
    """Data gradient of the 4x4 stride-2 discriminator block at full size: 4 exact parity launches == dilated gather."""
    from multimodal_segmentation_amd import _native as N
    Bc, Hi, Ci, Co = 8, 127, 64, 128
    Ho = (Hi - 4) // 2 + 1
    g = rnd(Bc, Ho, Ho, Co, seed=8).to(DEV)
    w = (rnd(4, 4, Ci, Co, seed=9) * 0.03).to(DEV)
    x = rnd(Bc, Hi, Hi, Ci, seed=10).to(DEV).requires_grad_(True)
    y = P.conv2d(x, w, None, stride=2, padding='valid')
    y.backward(g)                                               # parity path (Cout % 32 == 0)
    wf = torch.empty_like(w)
    N.call('mmseg_conv2d_wflip', w, wf, 4, 4, Ci, Co)
    dx = torch.empty(Bc, Hi, Hi, Ci, device=DEV)
    N.call('mmseg_conv2d_fwd', g, None, wf, None, None, dx, None, Bc, Ho, Ho, Co, 0, Hi, Hi, Ci, 4, 4, 2, 3, 3, 0, 1, 0, 0.0, 0)
    assert _rel(x.grad, dx) < 2e-5


def test_batchnorm_moments_and_softmax_round_invariants():
    x = (rnd(B, H, H, 64, seed=11) * 3 + 2).to(DEV)
    g, b = (torch.rand(64) + 0.5).to(DEV), rnd(64, seed=12).to(DEV)
    mm, mv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    y = P.batchnorm(x, g, b, mm, mv, True, relu=False)
    yc = y.reshape(-1, 64).double()
    assert (yc.mean(0).cpu() - b.cpu().double()).abs().max() < 1e-4          # mean = beta
    assert (yc.var(0, unbiased=False).sqrt().cpu() / g.cpu().double() - 1).abs().max() < 2e-3   # std = gamma (eps 1e-3)
    assert (mm.cpu() - 0.01 * x.reshape(-1, 64).mean(0).cpu()).abs().max() < 1e-5
    logits = (rnd(B, H, H, 8, seed=13) * 4).to(DEV)
    p, s = P.softmax_round(logits)
    assert (p.sum(-1) - 1).abs().max() < 1e-5
    assert bool(((s == 0) | (s == 1)).all()) and bool((s == torch.round(p)).all())
    assert bool((s.sum(-1) <= 1).all())                                        # at most one channel can exceed 0.5


def test_tps_identity_and_shift_at_full_size():
    from multimodal_segmentation_amd.layers.stn_spline import ThinPlateSpline2D
    tps = ThinPlateSpline2D((H, H), (5, 5), 8)
    vol = (rnd(B, H, H, 8, seed=14) > 0).float().to(DEV)
    out = tps([vol, torch.zeros(B, 25, 2, device=DEV)])
    assert float((out - vol).abs().max()) < 1e-5
    # a constant offset of every control point by k pixels (in normalised units) shifts the sampling grid by k pixels
    theta = torch.zeros(B, 25, 2, device=DEV)
    theta[..., 1] = 3.0 / (H - 1)                                              # +3 pixels along x
    out = tps([vol, theta])
    assert float((out[:, :, :-3] - vol[:, :, 3:]).abs().max()) < 1e-3
    assert float(out[:, :, -3:].abs().max()) < 1e-3                            # taps beyond the image contribute zero


def test_adam_closed_form_first_step():
    n = 1 << 20
    p0, g = rnd(n, seed=15).to(DEV), rnd(n, seed=16).to(DEV)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr_t = 1e-4 * math.sqrt(1 - 0.999) / (1 - 0.9)
    P.adam_step(p, g, m, v, lr_t)
    # first step: m = 0.1 g, v = 0.001 g^2  ->  delta = -lr_t * 0.1 g / (sqrt(0.001) |g| + 1e-7)
    ref = p0 - lr_t * 0.1 * g / (math.sqrt(0.001) * g.abs() + 1e-7)
    assert float((p - ref).abs().max()) < 5e-7                 # a few fp32 ulps of |p| ~ 1; the update itself is ~3e-5


def test_full_size_iteration_is_finite_and_learns_on_a_fixed_batch():
    """Generator-only DAFNet iterations at 256x256, batch 8 on ONE fixed batch (discriminators frozen, so this is plain
    minimisation): every loss finite and the weighted total decreases."""
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B)
    model = DAFNet(conf)
    model.build()
    d = Hh.make_step_data(B, H, H, seed=21)
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]
    rec = []
    for _ in range(4):
        h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
        vals = {k: h.history[k][0] for k in h.history.keys()}
        assert all(np.isfinite(v) for v in vals.values()), vals
        rec.append(vals['loss'])
    assert rec[-1] < rec[0], rec
    m = model.Segmentor.predict(model.Encoders_Anatomy[0].predict(d['x1']))
    assert m.shape == (B, H, H, 5) and abs(float(m.sum(-1).mean()) - 1) < 1e-4


def test_bitwise_reproducibility_of_a_discriminator_step_and_of_the_convolutions():
    """Every reduction runs in a fixed order (slabs, two-stage column sums) and the one scatter -- the TPS resampler's data
    gradient -- accumulates in 64-bit fixed point with integer atomics: the same step from the same state is bit-identical
    run to run, the whole generator step included."""
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    x, w = rnd(B, 128, 128, 128, seed=31).to(DEV), (rnd(3, 3, 128, 128, seed=32) * 0.03).to(DEV)
    cot = rnd(B, 128, 128, 128, seed=33).to(DEV)
    runs = []
    for _ in range(2):
        xd = x.clone().requires_grad_(True)
        wg = torch.zeros_like(w)
        y = P.conv2d(xd, w, None, wgrad=wg, anchor=nn.anchor(xd.device))
        y.backward(cot)
        runs.append((y.detach().clone(), xd.grad.clone(), wg.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    conf = Hh.make_conf(dafnet_config_chaos, 128, batch_size=4)
    model = DAFNet(conf)
    model.build()
    d = Hh.make_step_data(4, 128, 128, seed=3)
    w0 = model.D_Mask.get_weights()
    outs = []
    for _ in range(2):
        model.D_Mask.set_weights(w0)
        tr = model._d_trainer(model.D_Mask, 'D_Mask_trainer_rep', conf.d_mask_params.lr)
        tr.fit([d['dm_m1'], d['dm_m2']], [1.0, 0.0])
        outs.append((model.D_Mask.grad_arena.clone(), model.D_Mask.arena.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # a whole generator step (encoders, fuser incl. the TPS scatter, segmentor, decoder, frozen discriminators, Adam)
    gens = model._generator_models()
    g0 = [m.get_weights() for m in gens]
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]
    th = model.Anatomy_Fuser.params['theta/kernel']
    th.data.copy_(((rnd(*th.shape, seed=41)) * 0.002).to(DEV))           # a non-trivial warp, so that the scatter really scatters
    g0 = [m.get_weights() for m in gens]
    steps = []
    for _ in range(2):
        for m, w in zip(gens, g0):
            m.set_weights(w)
        tr = model.supervised_trainer
        tr.optimizer = nn.Adam(conf.lr)
        tr.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
        steps.append([m.grad_arena.clone() for m in gens] + [m.arena.clone() for m in gens])
    for a, b in zip(*steps):
        assert torch.equal(a, b)


def test_the_ctypes_stub_of_INTEGRATION_md_runs():
    """The hand-written binding shown in INTEGRATION.md (what a maintainer of the reference would add) is executed verbatim
    and checked against torch's convolution."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = [c for c in re.findall(r"```python\n(.*?)```", text, flags=re.S) if 'ctypes.CDLL' in c][0]
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                     # the snippet loads the library by its in-tree relative path
    try:
        exec(code, ns)
    finally:
        os.chdir(cwd)
    x = rnd(2, 24, 20, 16, seed=51).to(DEV)
    w = (rnd(3, 3, 16, 32, seed=52) * 0.1).to(DEV)
    b = rnd(32, seed=53).to(DEV)
    y = ns['conv3x3_same_relu'](x, w, b)
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1)).permute(0, 2, 3, 1)
    assert float((y - ref).abs().max()) < 1e-4


# ---- BASELINE.json configurations at their own sizes --------------------------------------------------------------------
def test_config2_generator_step_at_256_bs8_against_the_oracle():
    """BASELINE config #2 at its own size: one teacher-forced DAFNet-FiLM supervised generator step at 256 x 256, batch 8 -- all 20
    outputs, label maps, every loss term, BatchNorm moving statistics, gradients and post-Adam weights vs the oracle (run in
    fp32 here: the fp64 run needs > 60 GB; tolerances as in test_dafnet_step.py, gradients against a fixed bar because the fp32
    oracle IS the noise floor)."""
    from tests.test_dafnet_step import _generator_step_check
    _generator_step_check('film', H, True, 'cuda', B=B, oracle_dtype=torch.float32, out_tol=1e-3, grad_floor=6e-2)


def _fixed_batch_property_run(model, conf, d, ref_losses=None):
    """one generator step on a fixed batch: every loss term finite and -- reduced precision being a perturbation of the fp32
    arithmetic -- within 3e-2 (relative; 6e-2 for the discriminator heads, sums over 10^5 rounded features) of the fp32 step of
    the same weights on the same batch; then the same step twice from the same state is bit-identical"""
    gens = model._generator_models()
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]
    g0 = [m.get_weights() for m in gens]
    from tests import helpers as Hh
    with Hh.teacher_forcing(model, ref_losses['teacher'] if ref_losses else None):
        h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
    vals = {k: h.history[k][0] for k in h.history.keys()}
    assert all(np.isfinite(v) for v in vals.values()), vals
    if ref_losses is not None:
        for k, v in ref_losses['losses'].items():
            tol = 6e-2 if k in ('loss', 'D_Mask_loss', 'D_Image1_loss', 'D_Image2_loss') else 3e-2
            assert abs(vals[k] - v) <= tol * max(1.0, abs(v)), (k, v, vals[k])
        assert any(abs(vals[k] - v) > 1e-7 for k, v in ref_losses['losses'].items())     # and the 16-bit path really ran
    steps = []
    for _ in range(2):
        for m, w in zip(gens, g0):
            m.set_weights(w)
        model.supervised_trainer.optimizer = nn.Adam(conf.lr)
        model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
        steps.append([m.grad_arena.clone() for m in gens] + [m.arena.clone() for m in gens])
    for a, b in zip(*steps):
        assert torch.equal(a, b)


def test_config3_spade_bf16_at_256_bs8_properties():
    """BASELINE config #3's model, size, batch and dtype (DAFNet-SPADE, 256 x 256, batch 8, bf16 MFMA operands): full
    iterations of the executor's schedule stay finite, a fixed batch is learnt, and the step is bitwise reproducible.  (Parity
    against the oracle for this configuration: tests/test_dafnet_step.py::test_generator_step_reduced_precision... at 64 x 64.)"""
    from multimodal_segmentation_amd.configuration import dafnet_spade_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_spade_config_chaos, H, batch_size=B, compute_dtype='bf16')
    try:
        model = DAFNet(conf)
        model.build()
        assert P.set_conv_precision('bf16') == 'bf16'
        ex = DAFNetExecutor(conf, model)
        ex.init_train_data(slices_per_volume=4)
        losses = {n: [] for n in ex.get_loss_names()}
        for _ in range(2):
            ex.train_batch(losses)
        for k in ('supervised_Mask', 'adv_M', 'rec_X', 'adv_X1', 'adv_X2', 'KL', 'rec_Z', 'dis_M', 'dis_X1', 'dis_X2'):
            assert len(losses[k]) >= 2 and all(np.isfinite(float(v)) for v in losses[k]), (k, losses[k])
        # the fp32 step of a second model that carries the same weights, on the same batch, as the yardstick
        d = Hh.make_step_data(B, H, H, seed=23)
        all_models = lambda mm: mm._generator_models() + [mm.D_Mask, mm.D_Image1, mm.D_Image2]
        ref_model = DAFNet(Hh.make_conf(dafnet_spade_config_chaos, H, batch_size=B, compute_dtype='fp32'))
        ref_model.build()                                        # switches the library to fp32 MFMA
        for a, b in zip(all_models(ref_model), all_models(model)):
            a.set_weights(b.get_weights())
        tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]
        h = ref_model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
        ref = {'losses': {k: h.history[k][0] for k in h.history.keys()},
               'teacher': [ref_model.last_factors['s1'].detach().clone(), ref_model.last_factors['s2'].detach().clone()]}
        del ref_model
        assert P.set_conv_precision('bf16') == 'fp32'
        _fixed_batch_property_run(model, conf, d, ref)
    finally:
        P.set_conv_precision('fp32')


def test_config5_mmsdnet_three_modalities_fp16_at_320_bs16_properties():
    """BASELINE config #5's model, size, batch and dtype (3-modality MMSDNet -- the build-defined all-ordered-pairs extension --
    320 x 320, batch 16, fp16 MFMA operands with the static loss scale, fp32 master weights / gradients): full iterations of the
    executor's schedule stay finite, the fixed-batch generator objective decreases, gradients are finite (no fp16 overflow)."""
    from multimodal_segmentation_amd.configuration import mmsdnet3_config_chaos
    from multimodal_segmentation_amd.models.mmsdnet import MMSDNet
    from multimodal_segmentation_amd.model_executors.mmsdnet_executor import MMSDNetExecutor
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    S, Bb = 320, 16
    conf = Hh.make_conf(mmsdnet3_config_chaos, S, batch_size=Bb, compute_dtype='fp16')
    try:
        model = MMSDNet(conf)
        model.build()
        assert model.num_mod == 3 and model.n_out() == 15 and model.supervised_trainer.loss_scale == 1024.0
        ex = MMSDNetExecutor(conf, model)
        ex.init_train_data(slices_per_volume=8)
        losses = {n: [] for n in ex.get_loss_names()}
        ex.train_batch(losses)
        ex.train_batch(losses)
        for k in ('supervised_Mask', 'adv_M', 'rec_X', 'KL', 'rec_Z', 'dis_M'):
            assert len(losses[k]) == 2 and all(np.isfinite(float(v)) for v in losses[k]), (k, losses[k])
        for m in model._generator_models():
            assert bool(torch.isfinite(m.grad_arena).all()), m.name
        # fixed batch: the weighted generator objective decreases
        batch = next(ex.gen_labelled)
        x_list = [b for b in batch[:3]]
        m_list = [ex._five(b) for b in batch[3:]]
        tg = ex.generator_targets(x_list, m_list, True)
        eps = [np.random.RandomState(5 + i).standard_normal((Bb, 8)).astype(np.float32) for i in range(15)]
        rec = [model.supervised_trainer.fit(x_list, tg, eps=eps).history['loss'][0] for _ in range(3)]
        assert all(np.isfinite(v) for v in rec) and rec[-1] < rec[0], rec
    finally:
        P.set_conv_precision('fp32')


def _blocky_anatomy(rng, Bn, Hn, C=8):
    """piecewise-constant binary anatomies like the Rounding layer's output: thresholded smooth fields, one-hot over the channels"""
    from tests import helpers as Hh
    f = np.stack([Hh.smooth_field(rng, Bn, Hn, Hn, sigma=10.0)[..., 0] for _ in range(C)], -1)
    s = np.zeros_like(f)
    np.put_along_axis(s, f.argmax(-1)[..., None], 1.0, axis=-1)
    s[..., C - 1] = 0.0                     # some pixels without any anatomy channel, as after rounding
    return s.astype(np.float32)


@pytest.mark.parametrize('part', ['decoder', 'segmentor'])
def test_config3_subgraphs_at_256_against_the_operand_rounding_oracle(part):
    """BASELINE config #3 (DAFNet-SPADE, bf16 MFMA operands) AT ITS OWN IMAGE SIZE against the oracle, not against the product's own
    fp32 step: the two sub-graphs downstream of the rounded anatomies -- the SPADE decoder (30 convolutions, now with the fused gamma /
    beta launches) and the segmentor -- forward and backward on given (s, z) at 256 x 256, batch 2 (what the fp64 oracle affords in
    a test), with the oracle rounding the operands of exactly the convolutions the product multiplies in bf16.  Outputs within 2x the
    error of the same oracle run in fp32 (or 1e-2), every weight gradient within 1.5x its per-tensor noise floor (fp32 vs fp64
    operand-rounding oracle) or 5e-2."""
    from multimodal_segmentation_amd.configuration import dafnet_spade_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from oracle import models as OM
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    Bn = 2
    prev = O.set_conv_operand_rounding(torch.bfloat16)
    try:
        conf = Hh.make_conf(dafnet_spade_config_chaos, H, batch_size=Bn, compute_dtype='bf16')
        model = DAFNet(conf)
        model.build()
        rng = np.random.RandomState(11)
        s = _blocky_anatomy(rng, Bn, H)
        z = rng.standard_normal((Bn, 8)).astype(np.float32)
        comp, prefix = (model.Decoder, 'DEC/') if part == 'decoder' else (model.Segmentor, 'SEG/')
        R = rng.standard_normal((Bn, H, H, 1 if part == 'decoder' else 5)).astype(np.float32)
        res = {}
        for dt in (torch.float64, torch.float32):
            Pd = {k: v.clone().requires_grad_(not k.endswith(('moving_mean', 'moving_variance')))
                  for k, v in Hh.export_dafnet(model, dt).items() if k.startswith(prefix)}
            st, zt = torch.as_tensor(s, dtype=dt), torch.as_tensor(z, dtype=dt)
            yo = OM.decoder_spade(st, zt, Pd) if part == 'decoder' else OM.segmentor(st, Pd, True, [])
            (yo * torch.as_tensor(R, dtype=dt)).sum().backward()
            res[dt] = (yo.detach().double().numpy(), {k: v.grad.double().numpy() for k, v in Pd.items() if v.grad is not None})
        comp.zero_grad()
        with torch.enable_grad():
            sd, zd = nn.to_device(s, comp.device), nn.to_device(z, comp.device)
            yp = comp(sd, zd) if part == 'decoder' else comp(sd, training=True)
            torch.autograd.backward([yp], [nn.to_device(R, comp.device)])
        yo64, g64 = res[torch.float64]
        err = np.abs(yp.detach().float().cpu().numpy() - yo64).max()
        # the yardstick for the output: the SAME oracle run in fp32 -- where a pre-rounding value differs in its last fp32 bits an operand
        # rounds to the other bf16 neighbour, so two correct implementations decorrelate at the bf16 noise level through 30 convolutions
        floor_out = np.abs(res[torch.float32][0] - yo64).max()
        print('%s output: product err %.3e, fp32-oracle err %.3e (both vs the fp64 operand-rounding oracle)' % (part, err, floor_out))
        assert err <= max(2.0 * floor_out, 1e-2), '%s output: max abs err %.3e (oracle fp32 vs fp64: %.3e)' % (part, err, floor_out)
        pg = Hh.product_grads(model)
        worst = []
        for k, g in g64.items():
            nrm = np.linalg.norm(g)
            if nrm < 1e-12 or (k.endswith('/bias') and '_bn' not in k and part == 'segmentor' and not k.endswith('out/bias')):
                continue            # (segmentor: biases in front of a training-mode BatchNorm have an identically zero gradient)
            e = np.linalg.norm(pg[k] - g) / nrm
            floor = np.linalg.norm(res[torch.float32][1][k] - g) / nrm
            worst.append((e / max(1.5 * floor, 5e-2), e, floor, k))
        worst.sort(reverse=True)
        print('%s at 256 x 256 vs the operand-rounding oracle: output err %.3e; worst gradients (ratio, rel-L2, floor):' % (part, err), worst[:4])
        assert worst and worst[0][0] <= 1.0, worst[:4]
    finally:
        O.set_conv_operand_rounding(prev)
        P.set_conv_precision('fp32')


def test_config4_semi_supervised_lmix01_at_256_bs8_properties():
    """BASELINE config #4 at its own size (DAFNet, l_mix = 0.1 -> 1 labelled + 13 unlabelled volumes, 256 x 256, batch 8; the
    discriminators with their Spectral regularisers and the STN alignment path are part of every pass): an iteration is a supervised
    AND an unsupervised pass, each followed by both discriminator phases (dafnet_executor.py:380-387); separate Adam states; the
    unsupervised trainer has 18 outputs; everything finite; the iteration is bitwise reproducible.  (Parity of both trainers against the
    oracle: tests/test_dafnet_step.py at 64 x 64.)"""
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    finals = []
    for rep in range(2):
        conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, l_mix=0.1)
        model = DAFNet(conf)
        model.build()
        ex = DAFNetExecutor(conf, model)
        ex.init_train_data(slices_per_volume=8)
        assert len(model.unsupervised_trainer.specs) == 18 and len(model.supervised_trainer.specs) == 20
        losses = {n: [] for n in ex.get_loss_names()}
        ex.train_batch(losses)
        ex.train_batch(losses)
        assert model.supervised_trainer.optimizer.iterations == 2 and model.unsupervised_trainer.optimizer.iterations == 2
        assert model.D_Mask_trainer.optimizer.iterations == 8 and model.D_Image1_trainer.optimizer.iterations == 4
        assert len(losses['supervised_Mask']) == 4 and len(losses['dis_M']) == 8 and len(losses['dis_X1']) == 4
        for k in ('supervised_Mask', 'adv_M', 'rec_X', 'adv_X1', 'adv_X2', 'KL', 'rec_Z', 'dis_M', 'dis_X1', 'dis_X2'):
            assert all(np.isfinite(float(v)) for v in losses[k]), (k, losses[k])
        theta = model.Anatomy_Fuser.params['theta/kernel'].data
        assert float(theta.abs().max()) > 0, 'the STN head never received a gradient'
        for d_ in (model.D_Mask, model.D_Image1, model.D_Image2):
            assert bool(torch.isfinite(d_.arena).all())
        finals.append([m.arena.clone() for m in model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]])
        del model, ex
    for a, b in zip(*finals):
        assert torch.equal(a, b), 'two runs of the same two iterations differ'
