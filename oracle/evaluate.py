"""Oracle for the inference side of the hot path: validation losses, `predict_mask` fusion modes and the
per-volume test metrics / results.csv rows.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, over the oracle's parameter dict and `predict`-mode components
(BatchNorm with MOVING statistics):
  * MMSDNet.predict_mask                      models/mmsdnet.py:210-232            -> predict_mask
  * DAFNetExecutor.validate                   model_executors/dafnet_executor.py:303-355 -> validate_dafnet
  * MMSDNetExecutor.validate                  model_executors/mmsdnet_executor.py:205-236 -> validate_mmsdnet
  * ModelTester.test_modality_type            model_tester.py:45-79                -> test_rows / format_results
  * costs.dice (numpy metric)                 costs.py:31-41                       -> ops.dice_metric
`orc` is a DAFNetOracle or MMSDNetOracle: `orc.enc(x, mod)` is the anatomy encoder of modality `mod` in inference mode.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import models as M
from . import ops as O


def _t(x, like):
    return torch.as_tensor(np.asarray(x), dtype=like.dtype)


def _dtype_probe(orc):
    return next(iter(orc.P.values()))


@torch.no_grad()
def segment(orc, s):
    """Segmentor.predict: inference-mode BatchNorm (model_components/segmentor.py:9-29)"""
    return M.segmentor(s, orc.P, False, None)


@torch.no_grad()
def predict_mask(orc, modality_index, type, image_list, source_index=None):
    """models/mmsdnet.py:210-232.  idx2 = the modality to segment, idx1 = the other one (deformed onto idx2)."""
    assert type in ['simple', 'def', 'max', 'maxnostn']
    like = _dtype_probe(orc)
    idx2 = modality_index
    idx1 = (1 - idx2 if idx2 in (0, 1) else 0) if source_index is None else source_index
    s1 = orc.enc(_t(image_list[idx1], like), idx1)
    s2 = orc.enc(_t(image_list[idx2], like), idx2)
    if type == 'simple':
        return segment(orc, s2)
    if type == 'def':
        return segment(orc, M.anatomy_fuser(s1, s2, orc.P)[0])
    if type == 'max':
        return segment(orc, M.anatomy_fuser(s1, s2, orc.P)[1])
    return segment(orc, torch.maximum(s1, s2))            # 'maxnostn': np.max([s1, s2], axis=0)


def _dice_loss(masks, pred):
    return 1.0 - O.dice_metric(np.asarray(masks, np.float64), pred.double().numpy(), binarise=True)


@torch.no_grad()
def validate_dafnet(orc, x1, x2, m1, m2):
    """model_executors/dafnet_executor.py:303-355 -> the seven validation losses (1 - Dice, binarised predictions).
    NOTE the reference's naming: `s1_deformed, s2_fused = fuser([s1, s2])` -- the fusion that lands on modality 2."""
    like = _dtype_probe(orc)
    s1, s2 = orc.enc(_t(x1, like), 0), orc.enc(_t(x2, like), 1)
    s1_deformed, s2_fused = M.anatomy_fuser(s1, s2, orc.P)
    s2_deformed, s1_fused = M.anatomy_fuser(s2, s1, orc.P)
    out = OrderedDict()
    out['val_loss_mod1'] = _dice_loss(m1, segment(orc, s1))
    out['val_loss_mod2'] = _dice_loss(m2, segment(orc, s2))
    out['val_loss_mod2_mod1def'] = _dice_loss(m2, segment(orc, s1_deformed))
    out['val_loss_mod1_mod2def'] = _dice_loss(m1, segment(orc, s2_deformed))
    out['val_loss_mod2_fused'] = _dice_loss(m2, segment(orc, s2_fused))
    out['val_loss_mod1_fused'] = _dice_loss(m1, segment(orc, s1_fused))
    out['val_loss'] = float(np.mean([out['val_loss_mod1'], out['val_loss_mod2'], out['val_loss_mod2_mod1def'],
                                     out['val_loss_mod2_fused']]))            # dafnet_executor.py:354
    return out


@torch.no_grad()
def validate_mmsdnet(orc, x1, x2, m1, m2):
    """model_executors/mmsdnet_executor.py:205-236"""
    like = _dtype_probe(orc)
    s1, s2 = orc.enc(_t(x1, like), 0), orc.enc(_t(x2, like), 1)
    s1_deformed, s_fused = M.anatomy_fuser(s1, s2, orc.P)
    out = OrderedDict()
    out['val_loss_mod1'] = _dice_loss(m1, segment(orc, s1))
    out['val_loss_mod2'] = _dice_loss(m2, segment(orc, s2))
    out['val_loss_mod2_s1def'] = _dice_loss(m2, segment(orc, s1_deformed))
    out['val_loss_mod2_fused'] = _dice_loss(m2, segment(orc, s_fused))
    out['val_loss'] = float(np.mean([out['val_loss_mod1'], out['val_loss_mod2'], out['val_loss_mod2_s1def'],
                                     out['val_loss_mod2_fused']]))
    return out


def test_rows(orc, modality_index, type, volumes, num_masks):
    """model_tester.py:58-75.  volumes: iterable of (vol_id, [images of every modality], masks of `modality_index`)
    -> [(vol_id, joint Dice, [per-organ Dice])]"""
    rows = []
    for vol, images, mask in volumes:
        assert images[0].shape[0] > 0
        prd = predict_mask(orc, modality_index, type, images).double().numpy()
        mask = np.asarray(mask, np.float64)
        joint = O.dice_metric(mask, prd, binarise=True)
        sep = [O.dice_metric(mask[..., k:k + 1], prd[..., k:k + 1], binarise=True) for k in range(num_masks)]
        rows.append((vol, joint, sep))
    return rows


test_rows.__test__ = False      # not a pytest test


def format_results(rows, num_masks):
    """the text of results.csv (model_tester.py:57,71-73)"""
    lines = ['Vol, Dice, ' + ', '.join(['Dice%d' % k for k in range(num_masks)])]
    for vol, joint, sep in rows:
        lines.append(('%s, %.3f, ' + ', '.join(['%.3f'] * num_masks)) % ((str(vol), joint) + tuple(sep)))
    return '\n'.join(lines) + '\n'
