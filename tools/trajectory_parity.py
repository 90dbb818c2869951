#!/usr/bin/env python
"""The one-iteration map of the product (HIP kernels) against the oracle's at states DEEP in a training run: the oracle's own code
trains K iterations of tools/dice_seeds.py's schedule with its tensors on the GPU (torch-ROCm library kernels: fast enough for
hundreds of iterations), its complete state (weights, BatchNorm statistics, five Adam states) is transplanted into the product,
and iteration K + 1 is compared phase by phase, per tensor -- tests/test_free_running.py::synced_iteration, which the test-suite runs
at K = 0 / 3 / 8.

    python tools/trajectory_parity.py <seed> <K,K,...> [size=64] [batch=4]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from multimodal_segmentation_amd import nn, _native
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.loaders import synthetic
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from oracle import dafnet as OD
    from tests import helpers as Hh, test_free_running as FR
    seed = int(sys.argv[1])
    Ks = sorted(int(v) for v in sys.argv[2].split(','))
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    lr = 1e-3
    FR.LR = lr
    _native.load(); nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, lr=lr, seed=10 + seed)
    conf.d_mask_params['lr'] = lr; conf.d_image_params['lr'] = lr
    model = DAFNet(conf); model.build()
    odev = torch.device('cuda:0')
    orc = OD.DAFNetOracle({k: v.to(odev) for k, v in Hh.export_dafnet(model, torch.float32).items()}, dict(decoder_type='film', lr=lr, d_lr=lr))
    ex = DAFNetExecutor.__new__(DAFNetExecutor); ex.conf, ex.model = conf, model; ex.device = model.D_Mask.device
    train = synthetic.SyntheticPairedData(conf.input_shape, 4, list(range(6)), 8, 77)
    rng = np.random.RandomState(5 + seed)
    N = train.size()

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

    it = 0
    for K in Ks:
        while it < K:
            loss = orc.train_batch({k: v.to(odev) for k, v in Hh.to_torch(batch(), torch.float32).items()}, supervised=True)['supervised_Mask']
            it += 1
        lines = []
        d = batch()
        it += 1
        # generous bars: this is a measurement, the table is what counts
        big = (1.0, 1.0, 10.0, 1.0)
        FR.BARS_D, FR.BAR_M, FR.BAR_V, FR.BAR_W, FR.BAR_FLIP_FRAC, FR.BAR_BN = big, 1.0, 1.0, 10.0, 1.0, 1.0
        try:
            FR.synced_iteration(model, ex, orc, d, lines, 'seed %d, iteration %d of the oracle trajectory (seg loss %.4f)' % (seed, K + 1, loss), gen_bars=big)
        except AssertionError as exc:
            lines.append('ASSERTION: %r' % (exc,))
        for l in lines:
            if l.startswith('==') or l.startswith('   worst') or l.startswith('ASSERTION'):
                print(l[:420])
        s_mean = float(model.last_factors['s1'].detach().mean())
        print('   mean of the rounded anatomy s1: %.3f' % s_mean, flush=True)


if __name__ == '__main__':
    main()
