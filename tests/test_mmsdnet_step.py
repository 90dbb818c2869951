"""BASELINE config[0]: MMSDNet (mmsdnet_config_chaos) on 64x64 synthetic two-modality slices, batch 2 -- the product's
trainers against the oracle restatement with identical weights, inputs and random draws (teacher-forced at the Rounding
boundary, see test_dafnet_step.py).  The same iteration with THREE modalities (BASELINE config[4]'s model, a build-defined
extension: all ordered modality pairs, 60 outputs) is compared with the oracle's extension the same way."""
import numpy as np
import pytest
import torch

from multimodal_segmentation_amd import nn
from multimodal_segmentation_amd.configuration import mmsdnet_config_chaos, mmsdnet3_config_chaos
from multimodal_segmentation_amd.models.mmsdnet import MMSDNet
from oracle import mmsdnet as OM
from tests import helpers as Hh

TOL = 1e-3


@pytest.fixture(params=[pytest.param('cpu', id='cpu-standin'), pytest.param('cuda', marks=pytest.mark.gpu, id='mi355x')])
def device(request):
    if request.param == 'cpu':
        from tests import cpu_backend as cb
        cb.install()
        nn.set_default_device('cpu')
        yield 'cpu'
        cb.uninstall()
    else:
        assert torch.cuda.is_available()
        nn.set_default_device('cuda:0')
        yield 'cuda'


def _cmp(a, b, name, tol=TOL):
    err = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
    assert err <= tol, '%s: max abs err %.3e > %.1e' % (name, err, tol)


@pytest.mark.parametrize('nmod', [2, 3])
def test_mmsdnet_iteration(nmod, device):
    B, H = 2, 64
    conf = Hh.make_conf(mmsdnet_config_chaos if nmod == 2 else mmsdnet3_config_chaos, H)
    model = MMSDNet(conf)
    model.build()
    assert model.num_mod == nmod and model.n_out() == {2: 6, 3: 15}[nmod]
    n = model.n_out()
    rng = np.random.RandomState(3)
    th = model.Anatomy_Fuser.params['theta/kernel']
    th.data.copy_(torch.from_numpy((rng.standard_normal(th.shape) * 0.002).astype(np.float32)).to(th.data.device))
    orc = OM.MMSDNetOracle(Hh.export_mmsdnet(model, torch.float64), dict(lr=conf.lr, w_rec_X=conf.w_rec_X))

    d = Hh.make_step_data(B, H, H, seed=77)
    if nmod == 3:                       # a third modality: its own image, masks and discriminator-batch image
        r3 = np.random.RandomState(78)
        d['x3'], d['dm_x3'] = Hh.smooth_field(r3, B, H, H), Hh.smooth_field(r3, B, H, H)
        d['m3'] = Hh.add_residual(Hh.ellipse_masks(r3, B, H, H))
    t = Hh.to_torch(d, torch.float64)
    eps = [rng.standard_normal((B, 8)).astype(np.float32) for _ in range(n)]
    zs = [rng.standard_normal((B, 8)).astype(np.float32) for _ in range(n)]
    teps = [torch.as_tensor(e, dtype=torch.float64) for e in eps]
    xk = ['x%d' % (i + 1) for i in range(nmod)]
    mk = ['m%d' % (i + 1) for i in range(nmod)]

    # ---- generator fit (24 outputs; 60 with three modalities) ------------------------------------------------------
    ho = orc.generator_step_n([t[k] for k in xk], [t[k] for k in mk], teps, True)
    oo = orc.last_outputs
    teacher = [oo['s%d' % (i + 1)].detach().float().to(device) for i in range(nmod)]
    seg_t = [d[mk[j]] for j in model.seg_target_modalities(True)]          # 5 channels; Dice reads the first 4
    rec_t = [d[xk[j]] for j in model.rec_target_modalities()]
    if nmod == 2:       # the reference's target lists (mmsdnet_executor.py:254-258)
        assert [id(a) for a in seg_t] == [id(d[k]) for k in ('m1', 'm2', 'm2', 'm2', 'm1', 'm1')]
        assert [id(a) for a in rec_t] == [id(d[k]) for k in ('x1', 'x2', 'x2', 'x2', 'x1', 'x1')]
    # teacher-forced at the Rounding boundary through the same encoder hook as the DAFNet tests
    with Hh.teacher_forcing(model, teacher):
        h = model.supervised_trainer.fit([d[k] for k in xk], seg_t + [1.0] * n + rec_t + [0.0] * n, eps=eps)
    outs = model.supervised_trainer.last_outputs
    ref = oo['m_list'] + oo['adv_list'] + oo['rec_list'] + oo['kl_list']
    assert len(outs) == len(ref) == 4 * n
    for i, (a, b) in enumerate(zip(outs, ref)):
        _cmp(a.cpu().numpy(), b.detach().numpy(), 'output %d' % i)
        if i < n:      # label maps bit-exact wherever the oracle's decision is not a numerical tie (top-2 margin > 1e-4)
            rb = b.detach().numpy()
            top2 = np.sort(rb, axis=-1)[..., -2:]
            decided = (top2[..., 1] - top2[..., 0]) > 1e-4
            same = a.cpu().numpy().argmax(-1) == rb.argmax(-1)
            assert same[decided].all() and decided.mean() > 0.98, 'label map %d' % i
    for k, v in ho.items():
        rel = max(1.0, abs(v))
        _cmp(h.history[k][0] / rel, v / rel, 'loss ' + k)
    pg = Hh.product_grads_mmsdnet(model)
    for k, g in orc.last_grads.items():
        g = g.numpy()
        if np.abs(g).max() < 1e-7:
            continue
        err = np.linalg.norm(pg[k] - g) / max(np.linalg.norm(g), 1e-12)
        assert err <= 5e-2, 'grad %s: rel L2 %.3e' % (k, err)

    # ---- Z_Regressor fit on `predict`-mode anatomies -----------------------------------------------------------------
    s_list = orc.zreg_inputs(*[t[k] for k in xk])
    assert len(s_list) == n
    ro = orc.zreg_step(s_list, [torch.as_tensor(z, dtype=torch.float64) for z in zs])
    hz = model.Z_Regressor.fit([s.float().numpy() for s in s_list] + zs, zs)
    _cmp(hz.history['loss'][0], ro['loss'], 'rec_Z loss')

    # ---- D_Mask fit on the sampled fake pool -------------------------------------------------------------------------
    pool = orc.mask_pool(*[t['dm_' + k] for k in xk])
    assert pool.shape[0] == (3 * nmod - 2) * B
    idx = torch.as_tensor(rng.choice(pool.shape[0], B, replace=False))
    rd = orc.discriminator_step(t['dm_m1'], pool[idx])
    hd = model.D_Mask_trainer.fit([d['dm_m1'], pool[idx].float().numpy()], [1.0, 0.0])
    _cmp(hd.history['D_Mask_loss'][0], rd['D_Mask_loss'], 'dis_M')
    _cmp(hd.history['loss'][0], rd['loss'], 'dis_M total', 2e-3)


@pytest.mark.parametrize('nmod', [2, 3])
def test_mmsdnet_executor_schedule(nmod, device):
    """The executor's train_batch runs the three phases on device-resident synthetic data and keeps the loss names."""
    from multimodal_segmentation_amd.model_executors.mmsdnet_executor import MMSDNetExecutor
    conf = Hh.make_conf(mmsdnet_config_chaos if nmod == 2 else mmsdnet3_config_chaos, 64, batch_size=2)
    model = MMSDNet(conf)
    model.build()
    ex = MMSDNetExecutor(conf, model)
    ex.init_train_data(device_resident=(device == 'cuda'), slices_per_volume=1)
    losses = {n: [] for n in ex.get_loss_names()}
    ex.train_batch(losses)
    for k in ('supervised_Mask', 'adv_M', 'rec_X', 'KL', 'rec_Z', 'dis_M'):
        assert len(losses[k]) == 1 and np.isfinite(float(losses[k][0])), k
