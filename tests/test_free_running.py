"""FREE-RUNNING parity of whole DAFNet training iterations (no teacher forcing): the product's trainers, pools and optimisers
against the fp32 oracle from identical weights, batches and random draws, compared PER TENSOR on the complete training state
-- every weight, both Adam moments of every trainer, every BatchNorm moving statistic.

Why a second kind of model-level test beside tests/test_dafnet_step.py: the single teacher-forced generator step there cannot see
state that is carried BETWEEN fits (five Adam states, BatchNorm moving statistics read by the inference-mode pools, cached weight
images across optimiser steps) nor anything that only shows away from the initial weights (zero-initialised TPS head, empty
rounded anatomies, Adam at t = 1 where every update is +-lr).  Three cases:

  * `from_init`: one full iteration (generator fit, 2 D_Mask fits, D_Image1 / D_Image2 fits with their pools) from sharpened
    initial weights, phase by phase from synchronised states, then two more WITHOUT re-synchronisation under a stated bound;
  * `along_the_oracle_trajectory`: the oracle trains K iterations on its own, its COMPLETE state (weights, moving statistics,
    Adam t / m / v of all trainers) is transplanted into the product, and the next iteration is compared -- the one-step map of
    the two implementations at a state that training really visits (non-empty anatomies, non-zero warp, warm Adam).

What is compared and how tight.  Both sides compute in fp32 with different summation orders, so gradients agree to the fp32 noise
floor of this network (BatchNorm over few samples, ReLU / max-pool kinks on piecewise-constant anatomies), not to 1e-7:
  * Adam first moments m (linear in this step's gradient) and second moments v: relative L2 per tensor;
  * weights: Keras Adam moves a weight by lr_t * m / (sqrt(v) + eps), i.e. by +-lr where only this step's gradient is in the
    state, whatever its magnitude -- an element whose gradient is at the rounding level may take the opposite sign on the two
    sides.  Such elements are COUNTED (|dw| > lr/2) and bounded; everything else must agree to a fraction of lr;
  * BatchNorm moving statistics: relative to the tensor's max;
  * rounded anatomies of the generator step: flipped pixels counted (fp32-vs-fp32 differences of the softmax are ~1e-6, so a flip
    needs a pixel within that distance of 0.5).
The bars below are 3x what the MI355X run measured (profiles/r03_free_running_parity_*.txt holds the measured tables)."""
import os

import numpy as np
import pytest
import torch

from multimodal_segmentation_amd import nn
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from oracle import dafnet as OD
from tests import helpers as Hh

LR = 1e-3
# ---- bars (see the module docstring; measured values in profiles/r03_free_running_parity_*.txt) -------------------------------------
BAR_M = 2e-2          # Adam m: relative L2 per tensor (tensors whose oracle gradient is not at the noise level)
BAR_V = 4e-2          # Adam v: relative L2 per tensor
BAR_W = 0.25          # weights: max |dw| / lr over the decided elements (see compare_states)
BAR_FLIP_FRAC = 2e-3  # fraction of a tensor's DECIDED elements whose Adam update took the opposite sign (|dw| > lr / 2)
BAR_BN = 1e-4         # BatchNorm moving statistics: max |d| / max |stat|
BARS_D = (2e-3, 4e-3, 0.25, 2e-3)   # (m, v, w, flip fraction) of a discriminator fit: no BatchNorm, no Rounding -> a tight gradient


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


def _build(H, B, seed=10):
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, lr=LR, seed=seed)
    conf.d_mask_params['lr'] = LR
    conf.d_image_params['lr'] = LR
    model = DAFNet(conf)
    model.build()
    Hh.sharpen_anatomy_heads(model)
    ex = DAFNetExecutor.__new__(DAFNetExecutor)
    ex.conf, ex.model = conf, model
    ex.device = model.D_Mask.device
    orc = OD.DAFNetOracle(Hh.export_dafnet(model, torch.float32), dict(decoder_type='film', lr=LR, d_lr=LR))
    return conf, model, ex, orc


def _oracle_state(orc):
    W = {k: v.detach().double().cpu().numpy() for k, v in orc.P.items()}
    A = {key: (ad.t, {k: v.detach().double().cpu().numpy() for k, v in ad.m.items()},
               {k: v.detach().double().cpu().numpy() for k, v in ad.v.items()}) for key, ad in orc.adam.items()}
    return W, A


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _bias_before_bn(k):
    """the biases of the UNets' and the segmentor's Conv2D -> BatchNormalization blocks (every convolution of those models but the two
    1x1 softmax heads)"""
    return k.endswith('/bias') and k.startswith(('EA0/', 'EA1/', 'EAS/', 'SEG/')) and not k.endswith(('conv_anatomy/bias', 'out/bias'))


def compare_states(model, orc, lr_of, report=None, trainers=None):
    """per-tensor comparison of the complete training state -> dict of worst values; `report`: list collecting table lines"""
    Wp, Ap = Hh.product_state(model)
    Wo, Ao = _oracle_state(orc)
    assert set(Wp) == set(Wo)
    worst = dict(m=(0.0, ''), v=(0.0, ''), w=(0.0, ''), flip=(0.0, ''), bn=(0.0, ''))
    med = []
    lines = []
    for key, (t_o, mo, vo) in sorted(Ao.items()):
        if trainers is not None and key not in trainers:
            continue
        t_p, mp, vp = Ap[key]
        assert t_p == t_o, 'Adam iteration count of trainer %s: product %d, oracle %d' % (key, t_p, t_o)
        for k in sorted(mo):
            gm = np.abs(mo[k]).max()
            # a convolution bias in front of a training-mode BatchNorm has an identically zero gradient: the product does not
            # accumulate it (its Adam moments only decay and the bias keeps its value), the oracle holds rounding noise there
            if _bias_before_bn(k) or gm < 1e-9:
                continue
            em, ev = _rel_l2(mp[k], mo[k]), _rel_l2(vp[k], vo[k])
            lines.append('%-4s %-40s n=%8d  m %.2e  v %.2e' % (key, k, mo[k].size, em, ev))
            if em > worst['m'][0]:
                worst['m'] = (em, key + ':' + k)
            if ev > worst['v'][0]:
                worst['v'] = (ev, key + ':' + k)
    for k in sorted(Wo):
        if k.endswith(('moving_mean', 'moving_variance')):
            e = float(np.abs(Wp[k] - Wo[k]).max() / max(np.abs(Wo[k]).max(), 1e-12))
            if e > worst['bn'][0]:
                worst['bn'] = (e, k)
            continue
        if k.endswith('/u0'):
            assert np.array_equal(Wp[k], Wo[k]), k          # the Spectral regulariser's u is never assigned back (spectralnorm.py:234)
            continue
        lr = lr_of(k)
        d = np.abs(Wp[k] - Wo[k]) / lr
        if _bias_before_bn(k):
            continue        # random walk of a bias that the following BatchNorm removes (oracle) vs untouched (product): no observable
        # "decided" elements: first moment well above the tensor's noise level.  Keras Adam's update lr_t * m / (sqrt(v) + eps) is
        # sign-like for the others (exactly +-lr at t = 1, anything in between where |g| ~ eps = 1e-7), so a rounding-level
        # difference of the gradient moves them by up to 2 lr on either side: those are counted, not bounded
        mo = None
        for key in (trainers or sorted(Ao)):
            if k in Ao[key][1]:
                mo = np.abs(Ao[key][1][k])
        if mo is None:
            continue
        decided = mo > 0.1 * np.sqrt((mo * mo).mean())
        flips = d > 0.5
        und = int((flips & ~decided).sum())
        frac = float((flips & decided).sum()) / max(int(decided.sum()), 1)
        rest = float(d[decided & ~flips].max()) if (decided & ~flips).any() else 0.0
        med.append(float(np.median(d)))
        lines.append('W    %-40s n=%8d  decided %8d: max|dw|/lr %.4f, sign flips %d; undecided sign flips %d'
                     % (k, d.size, int(decided.sum()), rest, int((flips & decided).sum()), und))
        if rest > worst['w'][0]:
            worst['w'] = (rest, k)
        if frac > worst['flip'][0]:
            worst['flip'] = (frac, k)
    worst['median_dw'] = float(np.median(med)) if med else 0.0
    if report is not None:
        report.extend(lines)
    return worst


def _anatomy_flips(model, orc):
    n = 0
    for key in ('s1', 's2'):
        n += int((model.last_factors[key].detach().cpu().numpy() != orc.last_outputs[key].cpu().numpy()).sum())
    return n


def _write_report(name, device, lines):
    out = os.environ.get('MMSEG_PARITY_REPORT')
    if out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'free_running_parity_%s_%s.txt' % (name, device)), 'w') as f:
            f.write('\n'.join(lines) + '\n')


def _check(worst, flips, tag, bars=None):
    print(tag, 'anatomy flips', flips, {k: v for k, v in worst.items()})
    loose = 10.0 if flips else 1.0      # a flipped anatomy pixel is an O(1) local perturbation of everything downstream
    bm, bv, bw, bf = bars or (BAR_M, BAR_V, BAR_W, BAR_FLIP_FRAC)
    assert worst['m'][0] <= bm * loose, (tag, 'Adam m', worst['m'])
    assert worst['v'][0] <= bv * loose, (tag, 'Adam v', worst['v'])
    assert worst['bn'][0] <= BAR_BN * loose, (tag, 'BatchNorm moving statistic', worst['bn'])
    assert worst['flip'][0] <= bf * loose, (tag, 'sign-flipped Adam updates', worst['flip'])
    assert worst['w'][0] <= bw * loose, (tag, 'weights', worst['w'])


def _pool_diff(p, o):
    p, o = p.detach().float().cpu().numpy(), o.detach().float().cpu().numpy()
    return float((np.abs(p - o) > 1e-3).mean()), float(np.abs(p - o).max())


def synced_iteration(model, ex, orc, d, lines, tag, gen_bars=None):
    """One iteration of train_batch's schedule, phase by phase, with the oracle's complete state transplanted into the product
    BEFORE every phase: each of the five fits and both pool builds is then the same map applied to the same state on both sides
    (free-running inside the phase, Rounding included), and its result is compared per tensor.  Without the re-synchronisation the
    first Adam step (+-lr for every weight whatever its gradient's magnitude) turns rounding-level gradient differences into
    O(lr) weight differences that the next phase would be charged with."""
    dev = lambda a: nn.to_device(a, model.D_Mask.device)
    sel = lambda pool, idx: pool.index_select(0, torch.as_tensor(np.asarray(idx), dtype=torch.long, device=pool.device))
    odev = next(iter(orc.P.values())).device              # (the oracle's tensors may live on another device: tools/trajectory_parity.py)
    t = {k: v.to(odev) for k, v in Hh.to_torch(d, torch.float32).items()}

    def phase(name, keys, flips=0, bars=None):
        sub = []
        worst = compare_states(model, orc, lambda k: LR, sub, trainers=keys)
        lines.append('== %s, phase %s (flipped anatomy / pool pixels: %s)' % (tag, name, flips))
        lines.extend(sub)
        lines.append('   worst: %s' % worst)
        _check(worst, flips, tag + ' ' + name, bars)

    # generator fit
    Hh.load_oracle_state(model, orc)
    hp = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], Hh.dafnet_targets(d), eps=[d['eps1'], d['eps2']])
    ho = orc.generator_step(t['x1'], t['x2'], t['m1'], t['m2'], t['z1'], t['z2'], t['eps1'], t['eps2'], True)
    flips = _anatomy_flips(model, orc)
    for k, v in ho.items():
        assert abs(hp.history[k][0] - v) <= 1e-3 * max(1.0, abs(v)) * (10 if flips else 1), (tag, k, hp.history[k][0], v)
    phase('generator fit', ('sup',), flips, gen_bars)
    s_mean = float(model.last_factors['s1'].detach().mean())
    assert 0.01 < s_mean < 0.99, 'the rounded anatomies are trivial (mean %.3f): the free-running comparison would be vacuous' % s_mean
    # mask pools + the two D_Mask fits
    Hh.load_oracle_state(model, orc)
    p1, p2 = ex.mask_pools(dev(d['dm_x1']), dev(d['dm_x2']))
    o1, o2 = orc.mask_pools(t['dm_x1'], t['dm_x2'])
    pd = [_pool_diff(p1, o1), _pool_diff(p2, o2)]
    lines.append('== %s, mask pools: fraction of values off by > 1e-3: %.2e / %.2e, max %.2e / %.2e' % (tag, pd[0][0], pd[1][0], pd[0][1], pd[1][1]))
    assert max(pd[0][0], pd[1][0]) <= 1e-3, (tag, 'mask pools', pd)
    pool_flips = int(pd[0][0] > 0 or pd[1][0] > 0)
    model.D_Mask_trainer.fit([d['dm_m1'], sel(p1, d['dm_idx1'])], [1.0, 0.0])
    orc.discriminator_step('DM/', t['dm_m1'], o1[t['dm_idx1']])
    phase('D_Mask fit 1', ('DM',), pool_flips, BARS_D)
    Hh.load_oracle_state(model, orc)
    model.D_Mask_trainer.fit([d['dm_m2'], sel(p2, d['dm_idx2'])], [1.0, 0.0])
    orc.discriminator_step('DM/', t['dm_m2'], o2[t['dm_idx2']])
    phase('D_Mask fit 2', ('DM',), pool_flips, BARS_D)
    # image pools + D_Image1 / D_Image2 fits
    Hh.load_oracle_state(model, orc)
    y1, y2 = ex.image_pools(dev(d['di_x1']), dev(d['di_x2']), d['di_eps1'], d['di_eps2'])
    q1, q2 = orc.image_pools(t['di_x1'], t['di_x2'], t['di_eps1'], t['di_eps2'])
    pd = [_pool_diff(y1, q1), _pool_diff(y2, q2)]
    lines.append('== %s, image pools: fraction of values off by > 1e-3: %.2e / %.2e, max %.2e / %.2e' % (tag, pd[0][0], pd[1][0], pd[0][1], pd[1][1]))
    assert max(pd[0][0], pd[1][0]) <= 1e-3, (tag, 'image pools', pd)
    pool_flips = int(pd[0][0] > 0 or pd[1][0] > 0)
    model.D_Image1_trainer.fit([d['di_x1'], sel(y1, d['di_idx1'])], [1.0, 0.0])
    orc.discriminator_step('DI1/', t['di_x1'], q1[t['di_idx1']])
    phase('D_Image1 fit', ('DI1',), pool_flips, BARS_D)
    Hh.load_oracle_state(model, orc)
    model.D_Image2_trainer.fit([d['di_x2'], sel(y2, d['di_idx2'])], [1.0, 0.0])
    orc.discriminator_step('DI2/', t['di_x2'], q2[t['di_idx2']])
    phase('D_Image2 fit', ('DI2',), pool_flips, BARS_D)


def test_from_init(device):
    """iteration 1 phase by phase from synchronised states (tight, per tensor); iterations 2 and 3 of the same run free-running
    WITHOUT re-synchronisation: the per-tensor weight differences may grow by at most GROWTH per iteration (chaotic dynamics
    through Adam's sign-like first steps and the Rounding layer), and the losses stay together"""
    B, H = 2, (64 if device == 'cuda' else 48)          # (the CPU stand-in run checks the host logic: a smaller image keeps the suite short)
    conf, model, ex, orc = _build(H, B)
    lines = []
    try:
        synced_iteration(model, ex, orc, Hh.make_step_data(B, H, H, seed=1234), lines, 'iteration 1')
        Hh.load_oracle_state(model, orc)
        # measured (MI355X and the CPU stand-in alike): median |dw| / lr over the tensors 1e-4 after iteration 2, 1e-1 after iteration 3
        # -- Adam normalises every update to ~lr, so an element whose gradient is at the rounding level changes by O(lr) on one
        # side only; the stated bound is one decade above that, the losses stay within 2 %
        bound = {1: 2e-3, 2: 1.0}
        for it in ((1, 2) if device == 'cuda' else (1,)):
            d = Hh.make_step_data(B, H, H, seed=1234 + it)
            hp = Hh.product_train_batch(model, ex, d)
            ho = orc.train_batch(Hh.to_torch(d, torch.float32))
            worst = compare_states(model, orc, lambda k: LR)
            lines.append('== after free-running iteration %d (not re-synchronised): seg loss %.6f / %.6f, median |dw|/lr %.3e, worst %s'
                         % (it + 1, hp.history['Segmentor_loss'][0], ho['supervised_Mask'], worst['median_dw'], worst))
            assert abs(hp.history['Segmentor_loss'][0] - ho['supervised_Mask']) <= 2e-2 * max(1.0, abs(ho['supervised_Mask']))
            assert worst['median_dw'] <= bound[it], (it, worst['median_dw'])
    finally:
        _write_report('from_init', device, lines)


@pytest.mark.parametrize('K', [pytest.param(3, id='K3'), pytest.param(8, id='K8', marks=pytest.mark.gpu)])
def test_along_the_oracle_trajectory(K, device):
    """the one-iteration map of the two implementations at a state that training visits: the oracle trains K iterations alone, then
    iteration K + 1 is run phase by phase from the oracle's transplanted state (warm Adam moments, t > 1 in every trainer, moved
    BatchNorm statistics, a non-identity warp)"""
    B, H = 2, (64 if device == 'cuda' else 48)
    conf, model, ex, orc = _build(H, B, seed=11)
    for it in range(K):
        orc.train_batch(Hh.to_torch(Hh.make_step_data(B, H, H, seed=500 + it), torch.float32))
    Hh.load_oracle_state(model, orc)
    w0 = compare_states(model, orc, lambda k: LR)          # the transplant is exact
    assert w0['w'][0] == 0.0 and w0['m'][0] == 0.0 and w0['v'][0] == 0.0 and w0['bn'][0] == 0.0, w0
    assert float(model.Anatomy_Fuser.params['theta/kernel'].data.abs().max()) > 0
    lines = []
    try:
        # with warm Adam moments (10 % of m is this step's gradient) the generator fit agrees far more tightly than at t = 1: measured
        # on the MI355X 2.7e-3 (m) / 1.8e-3 (v) after K = 3 iterations, 3.4e-4 (m) / 3.5e-5 (v) after K = 8 (the same on the stand-in;
        # profiles/r03_free_running_parity_trajectory_K*.txt); bars = 3x / 6x those
        synced_iteration(model, ex, orc, Hh.make_step_data(B, H, H, seed=900), lines, 'iteration %d of the oracle trajectory' % (K + 1),
                         gen_bars=((2e-3, 4e-4) if K >= 8 else (8e-3, 6e-3)) + (BAR_W, BAR_FLIP_FRAC))
    finally:
        _write_report('trajectory_K%d' % K, device, lines)
