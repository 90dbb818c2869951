"""SURVEY 8(f) rank 1 against the oracle: `predict_mask` fusion modes (models/mmsdnet.py:210-232), the executors' `validate`
(dafnet_executor.py:303-355, mmsdnet_executor.py:205-236) and ModelTester's per-volume results.csv rows
(model_tester.py:45-79) -- the product's inference path (moving-statistics BatchNorm folded into the convolution launches,
device-side TPS warp and maximum) vs oracle/evaluate.py on identical weights and volumes.

Inference crosses TWO discontinuities (the encoders' Rounding and the metric's binarisation), so the comparison is made
(a) teacher-forced, component by component on the ORACLE's anatomies (softmax within 1e-3, label maps bit-exact wherever the
oracle's decision is not a numerical tie), and (b) free-running end to end (binarised masks agree on >= 99.9 % of the pixels,
every Dice value of results.csv / validate within 2e-3)."""
import os
import shutil

import numpy as np
import pytest
import torch

from multimodal_segmentation_amd import nn
from multimodal_segmentation_amd.configuration import dafnet_config_chaos, mmsdnet_config_chaos
from multimodal_segmentation_amd.loaders import synthetic
from oracle import dafnet as OD, mmsdnet as OMM, evaluate as OE, models as OMod
from tests import helpers as Hh

H = 64


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


def _liven(model, all_models):
    """Random initial weights give all-zero rounded anatomies and masks (softmax ~ 1/C everywhere): sharpen the two softmax
    heads, move the BatchNorm moving statistics away from (0, 1) and give the zero-initialised TPS head a small warp."""
    rng = np.random.RandomState(4)
    for m in all_models:
        for p in m.params.values():
            a = p.data.detach().cpu().numpy().copy()
            if p.name.endswith('moving_mean'):
                a += 0.05 * rng.standard_normal(a.shape).astype(np.float32)
            elif p.name.endswith('moving_variance'):
                a *= np.exp(0.2 * rng.standard_normal(a.shape)).astype(np.float32)
            elif p.name == 'conv_anatomy/kernel' or (p.name == 'out/kernel' and m.name == 'Segmentor'):
                a *= 40.0
            elif p.name == 'theta/kernel':
                a = (rng.standard_normal(a.shape) * 0.004).astype(np.float32)
            else:
                continue
            p.data.copy_(torch.from_numpy(a).to(p.data.device))
    from multimodal_segmentation_amd import ops
    ops.bump_weight_version()


def _memoise_encoders(orc):
    """the fp64 oracle UNet dominates the run time and the checks below encode the same volumes many times"""
    import hashlib
    enc, cache = orc.enc, {}

    def cached(x, mod, training=False, upd=None, soft_only=False):
        if training:
            return enc(x, mod, training, upd, soft_only)
        key = (mod, soft_only, hashlib.sha1(x.numpy().tobytes()).hexdigest())
        if key not in cache:
            cache[key] = enc(x, mod, False, None, soft_only)
        return cache[key]
    orc.enc = cached
    return orc


def _decided(ref, margin=1e-4):
    top2 = np.sort(ref, axis=-1)[..., -2:]
    return (top2[..., 1] - top2[..., 0]) > margin


def _check_components(model, orc, data, device):
    """teacher-forced: every inference component on the oracle's inputs"""
    like = next(iter(orc.P.values()))
    x = [torch.as_tensor(data.get_images_modi(i), dtype=like.dtype) for i in range(2)]
    s_o = []
    for i in range(2):
        soft_o = orc.enc(x[i], i, soft_only=True).numpy()
        s_p = model.Encoders_Anatomy[i].predict(data.get_images_modi(i))
        soft_p = model.Encoders_Anatomy[i].last_soft.detach().cpu().numpy()
        assert np.abs(soft_p - soft_o).max() < 1e-3, 'inference softmax of encoder %d' % i
        flips = int((s_p != np.round(soft_o)).sum())
        near = int((np.abs(soft_o - 0.5) < 1e-3).sum())
        assert flips <= near, 'encoder %d: %d flipped pixels, %d within 1e-3 of 0.5' % (i, flips, near)
        s_o.append(torch.as_tensor(np.round(soft_o), dtype=like.dtype))
        assert 0.02 < float(s_o[-1].mean()) < 0.5, 'degenerate anatomy: test would be trivial'
    for a, b in ((0, 1), (1, 0)):
        d_o, f_o = OMod.anatomy_fuser(s_o[a], s_o[b], orc.P)
        d_p, f_p = model.Anatomy_Fuser.predict([s_o[a].float().numpy(), s_o[b].float().numpy()])
        assert np.abs(d_p - d_o.numpy()).max() < 1e-3 and np.abs(f_p - f_o.numpy()).max() < 1e-3, 'fuser %d->%d' % (a, b)
        for name, s in (('own', s_o[b]), ('deformed', d_o), ('fused', f_o), ('maxnostn', torch.maximum(s_o[a], s_o[b]))):
            m_o = OE.segment(orc, s).numpy()
            m_p = model.Segmentor.predict(s.float().numpy())
            assert np.abs(m_p - m_o).max() < 1e-3, 'segmentor on %s' % name
            dec = _decided(m_o)
            assert (m_p.argmax(-1) == m_o.argmax(-1))[dec].all() and dec.mean() > 0.98, 'label map on %s' % name
            assert (np.round(m_o[..., :4]).sum() > 0), 'degenerate masks: test would be trivial'


def _volumes(data, modality_index):
    return [(v, [data.get_volume_images_modi(m, v) for m in range(2)], data.get_volume_masks_modi(modality_index, v))
            for v in data.volumes()]


def _check_predict_mask_and_rows(model, orc, data, conf):
    from multimodal_segmentation_amd import model_tester
    for mi in range(2):
        for mode in ('simple', 'def', 'max', 'maxnostn'):
            rows_o = OE.test_rows(orc, mi, mode, _volumes(data, mi), conf.num_masks)
            for (vol, images, mask), (_, joint_o, sep_o) in zip(_volumes(data, mi), rows_o):
                prd = model.predict_mask(mi, mode, images)
                prd_o = OE.predict_mask(orc, mi, mode, images).numpy()
                agree = (np.round(prd[..., :4]) == np.round(prd_o[..., :4])).mean()
                assert agree >= 0.999, 'predict_mask(%d, %s) vol %s: binarised masks agree on %.4f' % (mi, mode, vol, agree)
                joint, sep = model_tester.volume_scores(mask, prd, conf.num_masks)
                assert abs(joint - joint_o) < 2e-3 and np.abs(np.asarray(sep) - np.asarray(sep_o)).max() < 2e-3, (mi, mode, vol)
    with pytest.raises(AssertionError):
        model.predict_mask(0, 'nonsense', [data.get_images_modi(0), data.get_images_modi(1)])


def _check_results_csv(model, orc, data, conf):
    """ModelTester.run on the same volumes: folder / file layout, header, and every number of every results.csv"""
    from multimodal_segmentation_amd.model_tester import ModelTester
    shutil.rmtree(conf.folder, ignore_errors=True)
    tester = ModelTester(model, conf, test_data=data.copy())
    tester.run()
    rand = data.copy()
    rand.crop(conf.input_shape[:2])
    rand.randomise_pairs(length=2, seed=conf.seed)              # model_tester.py:41
    plain = data.copy()
    plain.crop(conf.input_shape[:2])
    n = 0
    for mi, mod in enumerate(conf.modality):
        for tag, dset in (('', plain), ('_rand', rand)):
            for mode in ('simple', 'def', 'max'):
                path = os.path.join(conf.folder, 'test_results_%s_%s_%s' % (conf.test_dataset, mod, mode + tag), 'results.csv')
                got = open(path).read().strip().split('\n')
                want = OE.format_results(OE.test_rows(orc, mi, mode, _volumes(dset, mi), conf.num_masks), conf.num_masks).strip().split('\n')
                assert got[0] == want[0] == 'Vol, Dice, Dice0, Dice1, Dice2, Dice3'
                assert len(got) == len(want) == 1 + len(data.volumes())
                for g, w in zip(got[1:], want[1:]):
                    g, w = g.split(', '), w.split(', ')
                    assert g[0] == w[0]
                    assert np.abs(np.asarray(g[1:], np.float64) - np.asarray(w[1:], np.float64)).max() <= 2e-3, (path, g, w)
                n += 1
    assert n == 12
    shutil.rmtree(conf.folder, ignore_errors=True)


def test_dafnet_inference_paths(device):
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=2, test_dataset='chaos', slices_per_volume=2)
    conf.folder = '/tmp/mmseg_test_eval_dafnet_' + device
    model = DAFNet(conf)
    model.build()
    _liven(model, model._generator_models())
    orc = _memoise_encoders(OD.DAFNetOracle(Hh.export_dafnet(model, torch.float64), dict(decoder_type=conf.decoder_type)))
    data = synthetic.SyntheticPairedData(conf.input_shape, conf.num_masks, [17, 18], 3, 4321)
    _check_components(model, orc, data, device)
    _check_predict_mask_and_rows(model, orc, data, conf)
    _check_results_csv(model, orc, data, conf)
    # validate: the executor evaluates its SWA clones (full-UNet clones of the shared-decoder encoders), = the live weights
    # before the first epoch end (dafnet_executor.py:303-355)
    ex = DAFNetExecutor(conf, model)
    ex.val_data = data.copy()
    losses = {k: [] for k in ex.get_loss_names()}
    ex.validate(losses)
    want = OE.validate_dafnet(orc, data.get_images_modi(0), data.get_images_modi(1), data.get_masks_modi(0), data.get_masks_modi(1))
    for k, v in want.items():
        assert len(losses[k]) == 1 and abs(float(losses[k][0]) - v) < 2e-3, (k, losses[k], v)
    assert 0.0 < want['val_loss'] < 1.0


def test_mmsdnet_inference_paths(device):
    from multimodal_segmentation_amd.models.mmsdnet import MMSDNet
    from multimodal_segmentation_amd.model_executors.mmsdnet_executor import MMSDNetExecutor
    conf = Hh.make_conf(mmsdnet_config_chaos, H, batch_size=2, test_dataset='chaos', slices_per_volume=2)
    conf.folder = '/tmp/mmseg_test_eval_mmsdnet_' + device
    shutil.rmtree(conf.folder, ignore_errors=True)
    model = MMSDNet(conf)
    model.build()
    _liven(model, model._generator_models())
    orc = _memoise_encoders(OMM.MMSDNetOracle(Hh.export_mmsdnet(model, torch.float64)))
    data = synthetic.SyntheticPairedData(conf.input_shape, conf.num_masks, [17, 18], 3, 4322)
    _check_components(model, orc, data, device)
    _check_predict_mask_and_rows(model, orc, data, conf)
    ex = MMSDNetExecutor(conf, model)
    ex.val_data = data.copy()
    losses = {k: [] for k in ex.get_loss_names()}
    ex.validate(losses)                                        # live models (mmsdnet_executor.py:205-236)
    want = OE.validate_mmsdnet(orc, data.get_images_modi(0), data.get_images_modi(1), data.get_masks_modi(0), data.get_masks_modi(1))
    for k, v in want.items():
        assert len(losses[k]) == 1 and abs(float(losses[k][0]) - v) < 2e-3, (k, losses[k], v)
