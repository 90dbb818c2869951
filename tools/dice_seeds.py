#!/usr/bin/env python
"""Dice criterion of the north_star ("Dice within +-0.002 of the reference on a fixed synthetic split"), made decidable as far as
the available compute allows: SEVERAL seeds x {product on the GPU, oracle on the CPU}, each trained free-running (no teacher
forcing) from identically seeded weights on identical batches and draws, evaluated as the reference evaluates -- on the
stochastic-weight-averaged model (callbacks/swa.py: running mean of the weights, BatchNorm moving statistics included) -- on a
fixed synthetic validation split of 256 paired slices.

    python tools/dice_seeds.py <seed> <product|oracle|standin|oracle_gpu> [iterations=500] [size=64] [batch=4] [lr=1e-3] [swa_from=350] [swa_every=10]
        product: the HIP kernels on the GPU; oracle: oracle/ on the CPU; standin: the PRODUCT's host logic (graphs, trainers, pools,
        caches) on tests/cpu_backend.py, i.e. torch-CPU arithmetic under the product's Python -- separates "host logic" from "kernel
        numerics" without a GPU.  oracle_gpu: the ORACLE's own code (oracle/*.py, plain torch ops) with its tensors on the GPU, i.e. on
        torch-ROCm's library kernels (MIOpen / rocBLAS) -- a third implementation of the same arithmetic that owes nothing to csrc/, and
        fast enough for many seeds.  DICE_LABEL=<name> relabels the RESULT line (e.g. a second oracle realisation with ORACLE_THREADS=1)
    python tools/dice_seeds.py summary <log> [<log> ...]        # mean +- 95 % CI of (product - oracle) over the seeds
    DICE_CHECKS=10,25,50,100,200 python tools/dice_seeds.py ...  # also evaluate the LIVE model after these iterations (CHECK lines)

One line `RESULT seed side dice_mod1 dice_mod2 dice_mean` is printed at the end of a run; `summary` pairs the lines by seed.
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def trajectory(paths):
    """|Dice(product) - Dice(oracle)| of the live model at the check points (DICE_CHECKS): the two trajectories start from the
    same weights and see the same batches, so early check points test the implementation; the chaotic growth of the fp32
    rounding differences shows up as the gap widening with the iteration count"""
    chk = {}
    for p in paths:
        for line in open(p):
            if line.startswith('CHECK'):
                _, seed, side, it, d1, d2, dm = line.split()
                chk.setdefault(int(it), {}).setdefault(int(seed), {})[side] = (float(d1), float(d2), float(dm))
    if not chk:
        return
    print('iteration   seeds   mean Dice product / oracle    mean |delta| (per modality)   max |delta|')
    for it in sorted(chk):
        rows = [v for v in chk[it].values() if 'product' in v and 'oracle' in v]
        if not rows:
            continue
        d = np.array([[abs(r['product'][0] - r['oracle'][0]), abs(r['product'][1] - r['oracle'][1])] for r in rows])
        print('%9d   %5d   %.4f / %.4f               %.5f                       %.5f' % (
            it, len(rows), np.mean([r['product'][2] for r in rows]), np.mean([r['oracle'][2] for r in rows]), d.mean(), d.max()))


def weight_trajectory(paths):
    """relative difference of the weight norm (segmentation path) between the two sides after n iterations: starts at the fp32
    rounding level if one step is the same computation, and grows as the chaotic training dynamics amplify it"""
    chk = {}
    for p in paths:
        for line in open(p):
            if line.startswith('WCHECK'):
                _, seed, side, it, n, nrm, sm = line.split()
                chk.setdefault(int(it), {}).setdefault(int(seed), {})[side] = (float(nrm), float(sm), int(n))
    if not chk:
        return
    print('iteration   seeds   |norm_p - norm_o| / norm_o (per seed)          |sum_p - sum_o| / norm_o (per seed)')
    for it in sorted(chk):
        rows = [v for _, v in sorted(chk[it].items()) if 'product' in v and 'oracle' in v]
        if not rows:
            continue
        assert all(r['product'][2] == r['oracle'][2] for r in rows), 'the two sides count different numbers of weights'
        print('%9d   %5d   %-45s  %s' % (it, len(rows), ' '.join('%.2e' % (abs(r['product'][0] - r['oracle'][0]) / r['oracle'][0]) for r in rows),
                                          ' '.join('%.2e' % (abs(r['product'][1] - r['oracle'][1]) / r['oracle'][0]) for r in rows)))


def summary(paths):
    weight_trajectory(paths)
    trajectory(paths)
    res = {}
    for p in paths:
        for line in open(p):
            if line.startswith('RESULT'):
                _, seed, side, d1, d2, dm = line.split()
                res.setdefault(int(seed), {})[side] = (float(d1), float(d2), float(dm))
    seeds = sorted(s for s, v in res.items() if 'product' in v and 'oracle' in v)
    if not seeds:
        print('no seed has both sides yet:', {s: list(v) for s, v in res.items()})
        return
    print('seed   product (mod1 mod2 mean)        oracle (mod1 mod2 mean)         product - oracle (mean)')
    diffs, pm, om = [], [], []
    for s in seeds:
        p, o = res[s]['product'], res[s]['oracle']
        diffs.append(p[2] - o[2]); pm.append(p[2]); om.append(o[2])
        print('%4d   %.4f %.4f %.4f            %.4f %.4f %.4f            %+.4f' % ((s,) + p + o + (p[2] - o[2],)))
    n = len(diffs)
    mean = float(np.mean(diffs))
    sd = float(np.std(diffs, ddof=1)) if n > 1 else float('nan')
    tcrit = {2: 12.71, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365, 9: 2.306, 10: 2.262, 11: 2.228, 12: 2.201,
             13: 2.179, 14: 2.160, 15: 2.145, 16: 2.131, 17: 2.120, 18: 2.110, 19: 2.101, 20: 2.093, 21: 2.086, 22: 2.080, 23: 2.074,
             24: 2.069, 25: 2.064, 26: 2.060, 27: 2.056, 28: 2.052, 29: 2.048, 30: 2.045}.get(n, 1.96)     # Student t, 97.5 %, n - 1 degrees of freedom
    half = tcrit * sd / math.sqrt(n) if n > 1 else float('nan')
    print('n = %d seeds: Dice product %.4f +- %.4f (sd), oracle %.4f +- %.4f (sd)' % (n, np.mean(pm), np.std(pm, ddof=1) if n > 1 else 0,
                                                                                     np.mean(om), np.std(om, ddof=1) if n > 1 else 0))
    print('paired difference product - oracle: mean %+.4f, sd %.4f, 95 %% CI [%+.4f, %+.4f]' % (mean, sd, mean - half, mean + half))
    inside = (mean - half) <= 0.002 and (mean + half) >= -0.002
    print('|delta| <= 0.002 is %s the 95 %% CI (half width %.4f): the criterion is %s at this sample size'
          % ('compatible with' if inside else 'outside', half, 'not rejected' if inside else 'rejected'))


def main():
    if sys.argv[1] == 'summary':
        return summary(sys.argv[2:])
    import torch
    from multimodal_segmentation_amd import nn, _native, costs
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.loaders import synthetic
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from oracle import dafnet as OD, models as OM
    from tests import helpers as Hh

    seed, side = int(sys.argv[1]), sys.argv[2]
    arg = lambda i, d, t: t(sys.argv[i]) if len(sys.argv) > i else d
    iters, H, B, lr = arg(3, 500, int), arg(4, 64, int), arg(5, 4, int), arg(6, 1e-3, float)
    swa_from, swa_every = arg(7, 350, int), arg(8, 10, int)
    checks = set(int(v) for v in os.environ.get('DICE_CHECKS', '').split(',') if v)     # iterations after which the live model is evaluated
    wchecks = set(int(v) for v in os.environ.get('DICE_WCHECKS', '').split(',') if v)   # ... after which the weight norm is printed
    odt = torch.float32                                   # the oracle runs in fp32 here (CPU time); the product is fp32 too
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', max(1, min(len(os.sched_getaffinity(0)), 16)))))
    label = os.environ.get('DICE_LABEL', side)
    odev = torch.device('cuda:0' if side == 'oracle_gpu' else 'cpu')        # where the oracle's tensors live
    if side == 'oracle_gpu':
        side = 'oracle'
    if side in ('oracle', 'standin'):                     # no GPU: build the identically seeded model on the CPU stand-in (oracle: to export weights)
        from tests import cpu_backend as _cb
        _cb.install(); nn.set_default_device('cpu')
    else:
        _native.load(); nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, lr=lr, seed=10 + seed)
    conf.d_mask_params['lr'] = lr; conf.d_image_params['lr'] = lr
    model = DAFNet(conf); model.build()
    orc = OD.DAFNetOracle({k: v.to(odev) for k, v in Hh.export_dafnet(model, odt).items()},
                          dict(decoder_type='film', lr=lr, d_lr=lr)) if side == 'oracle' else None
    ex = DAFNetExecutor.__new__(DAFNetExecutor); ex.conf, ex.model = conf, model; ex.device = model.D_Mask.device
    train = synthetic.SyntheticPairedData(conf.input_shape, 4, list(range(6)), 8, 77)             # fixed split for every seed
    val = synthetic.SyntheticPairedData(conf.input_shape, 4, list(range(14, 30)), 16, 78)          # 16 volumes x 16 = 256 slices
    rng = np.random.RandomState(5 + seed)
    N = train.size()
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

    seg_models = lambda: [model.Encoders_Anatomy[0], model.Encoders_Anatomy[1], model.Segmentor]
    swa, n_swa = None, 0

    def swa_update():
        """running mean of the weights incl. BatchNorm moving statistics (callbacks/swa.py:29-40)"""
        nonlocal swa, n_swa
        if side == 'oracle':
            cur = {k: v.detach().clone() for k, v in orc.P.items() if k.startswith(('EA0/', 'EA1/', 'EAS/', 'SEG/'))}
        else:
            cur = {'%d/%d' % (i, j): w for i, m in enumerate(seg_models()) for j, w in enumerate(m.get_weights())}
        if swa is None:
            swa, n_swa = cur, 1
        else:
            for k in swa:
                swa[k] = (swa[k] * n_swa + cur[k]) / (n_swa + 1)
            n_swa += 1

    def evaluate_live():
        """Dice of the LIVE weights (inference-mode BatchNorm) on the validation split: the trajectory check points"""
        x1, x2 = val.get_images_modi(0), val.get_images_modi(1)
        m1, m2 = val.get_masks_modi(0), val.get_masks_modi(1)
        out = []
        for i, (x, m) in enumerate(((x1, m1), (x2, m2))):
            if side == 'oracle':
                with torch.no_grad():
                    preds = [OM.segmentor(orc.enc(torch.as_tensor(x[j:j + 32], dtype=odt).to(odev), i), orc.P, False, None).cpu().numpy()
                             for j in range(0, len(x), 32)]
            else:
                preds = [model.Segmentor.predict(model.Encoders_Anatomy[i].predict(x[j:j + 32])) for j in range(0, len(x), 32)]
            out.append(costs.dice(m, np.concatenate(preds, 0), binarise=True))
        return out

    def evaluate_swa():
        x1, x2 = val.get_images_modi(0), val.get_images_modi(1)
        m1, m2 = val.get_masks_modi(0), val.get_masks_modi(1)
        out = []
        if side == 'oracle':
            P = dict(orc.P); P.update(swa)
            o2 = OD.DAFNetOracle(P, dict(orc.conf))
            with torch.no_grad():
                for i, (x, m) in enumerate(((x1, m1), (x2, m2))):
                    preds = [OM.segmentor(o2.enc(torch.as_tensor(x[j:j + 32], dtype=odt).to(odev), i), P, False, None).cpu().numpy() for j in range(0, len(x), 32)]
                    out.append(costs.dice(m, np.concatenate(preds, 0), binarise=True))
        else:
            live = [m.get_weights() for m in seg_models()]
            for i, m in enumerate(seg_models()):
                m.set_weights([np.asarray(swa['%d/%d' % (i, j)], np.float32) for j in range(len(live[i]))])
            for i, (x, m) in enumerate(((x1, m1), (x2, m2))):
                preds = [model.Segmentor.predict(model.Encoders_Anatomy[i].predict(x[j:j + 32])) for j in range(0, len(x), 32)]
                out.append(costs.dice(m, np.concatenate(preds, 0), binarise=True))
            for m, w in zip(seg_models(), live):
                m.set_weights(w)
        return out

    t0 = time.time()
    for it in range(iters):
        d = batch()
        if side in ('product', 'standin'):
            loss = product_step(d)
        else:
            loss = orc.train_batch({k: v.to(odev) for k, v in Hh.to_torch(d, odt).items()}, supervised=True)['supervised_Mask']
        if it >= swa_from and (it - swa_from) % swa_every == 0:
            swa_update()
        if it % 50 == 0 or it == iters - 1:
            print('seed %d %s iter %4d seg loss %.4f (%.0f s)' % (seed, label, it, loss, time.time() - t0), flush=True)
        if it + 1 in checks:
            c1, c2 = evaluate_live()
            print('CHECK %d %s %d %.5f %.5f %.5f' % (seed, label, it + 1, c1, c2, 0.5 * (c1 + c2)), flush=True)
        if it + 1 in wchecks:
            # a well-conditioned trajectory observable: L2 norm and sum of the weights of the segmentation path (float64 on the host).
            # Kernels, gamma / beta and moving variances only: a convolution bias in front of a BatchNorm has an identically zero
            # gradient (the product skips it, the reference and the oracle feed rounding noise through Adam into a random walk of
            # that bias, which the BatchNorm -- and its moving mean -- absorbs without any effect on the output)
            keep = lambda k: k.startswith(('EA0/', 'EA1/', 'EAS/', 'SEG/')) and k.endswith(('/kernel', '/gamma', '/beta', '/moving_variance'))
            if side == 'oracle':
                ws = [v.detach().double().cpu().numpy().ravel() for k, v in sorted(orc.P.items()) if keep(k)]
            else:           # the same tensors under the oracle's names (the encoders' shared up-path counted once)
                ws = [v.detach().double().cpu().numpy().ravel() for k, v in sorted(Hh.export_dafnet(model, torch.float32).items()) if keep(k)]
            w = np.concatenate(ws)
            print('WCHECK %d %s %d %d %.12e %.12e' % (seed, label, it + 1, w.size, np.sqrt((w * w).sum()), w.sum()), flush=True)
    if swa is None:
        return
    d1, d2 = evaluate_swa()
    print('RESULT %d %s %.5f %.5f %.5f' % (seed, label, d1, d2, 0.5 * (d1 + d2)), flush=True)


if __name__ == '__main__':
    main()
