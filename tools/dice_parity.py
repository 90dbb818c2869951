#!/usr/bin/env python
"""Free-running Dice parity: the product (HIP kernels on the GPU) and the oracle (torch-CPU) train from identical weights on
identical batches with identical random draws (no teacher forcing), then both are evaluated on a fixed synthetic validation
split.  The Rounding layer makes the two trajectories diverge pixel by pixel, so this is a statistical statement, the one the
north_star asks for: |Dice_product - Dice_oracle| on a fixed synthetic split.

    python tools/dice_parity.py [iterations=60] [size=64] [batch=4] [lr=1e-3] [both|product|oracle] [eval_every=10] [fp32|bf16|fp16]

`product` / `oracle` run one side only (same seeds, hence the same initial weights, batches and draws): the oracle side needs
no GPU and can run for hours elsewhere; compare the printed Dice trajectories afterwards.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multimodal_segmentation_amd import nn, _native, costs
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.loaders import synthetic
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from oracle import dafnet as OD, models as OM
from tests import helpers as Hh

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
LR = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
SIDE = sys.argv[5] if len(sys.argv) > 5 else 'both'
EVERY = int(sys.argv[6]) if len(sys.argv) > 6 else 10
DTYPE = sys.argv[7] if len(sys.argv) > 7 else 'fp32'          # product side only: fp32 | bf16 | fp16
torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
if SIDE == 'oracle':                      # no GPU: build the (identically seeded) model on the CPU stand-in just to export weights
    from tests import cpu_backend as _cb
    _cb.install(); nn.set_default_device('cpu')
else:
    _native.load(); nn.set_default_device('cuda:0')
conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, lr=LR, compute_dtype=DTYPE if SIDE != 'oracle' else 'fp32')
conf.d_mask_params['lr'] = LR; conf.d_image_params['lr'] = LR
model = DAFNet(conf); model.build()
orc = OD.DAFNetOracle(Hh.export_dafnet(model, torch.float64), dict(decoder_type='film', lr=LR, d_lr=LR))
ex = DAFNetExecutor.__new__(DAFNetExecutor); ex.conf, ex.model = conf, model; ex.device = model.D_Mask.device
train = synthetic.SyntheticPairedData(conf.input_shape, 4, list(range(6)), 8, 77)
val = synthetic.SyntheticPairedData(conf.input_shape, 4, [14, 15], 8, 78)
rng = np.random.RandomState(5)
N = train.size()
T = lambda a: torch.as_tensor(a, dtype=torch.float64)
dev = lambda a: nn.to_device(a, ex.device)


def batch():
    d = {}
    for pre in ('', 'dm_', 'di_'):
        idx = rng.choice(N, B, replace=False)
        d[pre + 'x1'], d[pre + 'x2'] = train.get_images_modi(0)[idx], train.get_images_modi(1)[idx]
        if pre == '':
            d['m1'], d['m2'] = Hh.add_residual(train.get_masks_modi(0)[idx]), Hh.add_residual(train.get_masks_modi(1)[idx])
    d['dm_m1'] = train.get_masks_modi(0)[rng.choice(N, B, replace=False)]
    d['dm_m2'] = train.get_masks_modi(1)[rng.choice(N, B, replace=False)]
    for k in ('z1', 'z2', 'eps1', 'eps2', 'di_eps1', 'di_eps2'):
        d[k] = rng.standard_normal((B, 8)).astype(np.float32)
    d['dm_idx1'], d['dm_idx2'] = rng.choice(2 * B, B, replace=False), rng.choice(2 * B, B, replace=False)
    d['di_idx1'], d['di_idx2'] = rng.choice(3 * B, B, replace=False), rng.choice(3 * B, B, replace=False)
    return d


def product_step(d):
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]
    h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
    p1, p2 = ex.mask_pools(dev(d['dm_x1']), dev(d['dm_x2']))
    sel = lambda pool, idx: pool.index_select(0, torch.as_tensor(idx, dtype=torch.long, device=pool.device))
    model.D_Mask_trainer.fit([d['dm_m1'], sel(p1, d['dm_idx1'])], [1.0, 0.0])
    model.D_Mask_trainer.fit([d['dm_m2'], sel(p2, d['dm_idx2'])], [1.0, 0.0])
    y1, y2 = ex.image_pools(dev(d['di_x1']), dev(d['di_x2']), d['di_eps1'], d['di_eps2'])
    model.D_Image1_trainer.fit([d['di_x1'], sel(y1, d['di_idx1'])], [1.0, 0.0])
    model.D_Image2_trainer.fit([d['di_x2'], sel(y2, d['di_idx2'])], [1.0, 0.0])
    return h.history['Segmentor_loss'][0]


def dice_of(masks_pred, masks_true):
    return costs.dice(masks_true, masks_pred, binarise=True)


def evaluate():
    x1, x2 = val.get_images_modi(0), val.get_images_modi(1)
    m1, m2 = val.get_masks_modi(0), val.get_masks_modi(1)
    dp_, do_ = [float('nan')] * 2, [float('nan')] * 2
    if SIDE != 'oracle':
        pp = [model.Segmentor.predict(model.Encoders_Anatomy[i].predict(x)) for i, x in enumerate((x1, x2))]
        dp_ = [dice_of(pp[0], m1), dice_of(pp[1], m2)]
    if SIDE != 'product':
        with torch.no_grad():
            po = [OM.segmentor(orc.enc(T(x), i), orc.P, False, None).numpy() for i, x in enumerate((x1, x2))]
        do_ = [dice_of(po[0], m1), dice_of(po[1], m2)]
    return dp_, do_


t0 = time.time()
for it in range(ITERS):
    d = batch()
    lp = product_step(d) if SIDE != 'oracle' else float('nan')
    lo = orc.train_batch(Hh.to_torch(d, torch.float64), supervised=True)['supervised_Mask'] if SIDE != 'product' else float('nan')
    if it % EVERY == 0 or it == ITERS - 1:
        dp_, do_ = evaluate()
        print('iter %3d  seg loss product %.4f oracle %.4f | val Dice product %.4f %.4f  oracle %.4f %.4f  |diff| %.4f %.4f  (%.0f s)'
              % (it, lp, lo, dp_[0], dp_[1], do_[0], do_[1], abs(dp_[0] - do_[0]), abs(dp_[1] - do_[1]), time.time() - t0), flush=True)
