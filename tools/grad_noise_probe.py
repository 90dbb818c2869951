#!/usr/bin/env python
"""One free-running supervised generator step, product (GPU, fp32) and oracle (CPU, fp32) from the same weights and draws: how large
are the gradient components that are ~0, relative to Keras Adam's epsilon (1e-7)?  For every generator tensor: the fraction of
elements with |g| < 1e-7 on either side and, among the elements where the fp64 oracle's gradient is < 1e-9 in magnitude ("true
zeros"), the rms and max of the fp32 gradient each implementation computes.  MMSEG_BN_BIAS_GRAD=1 includes the biases in front of
BatchNorms.      python tools/grad_noise_probe.py [size=64] [batch=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from multimodal_segmentation_amd import nn, _native
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from oracle import dafnet as OD
    from tests import helpers as Hh
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    _native.load(); nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=B, lr=1e-3, seed=10)
    model = DAFNet(conf); model.build()
    d = Hh.make_step_data(B, H, H, seed=3)
    B1 = np.ones((B, 1), np.float32)
    tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [B1] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [B1] * 4 + [np.zeros(B, np.float32)] * 2 + [d['z1'], d['z2']]
    grads = {}
    for name, dt in (('oracle fp64', torch.float64), ('oracle fp32', torch.float32)):
        orc = OD.DAFNetOracle(Hh.export_dafnet(model, dt), dict(decoder_type='film', lr=1e-3, d_lr=1e-3))
        t = Hh.to_torch(d, dt)
        orc.generator_step(t['x1'], t['x2'], t['m1'], t['m2'], t['z1'], t['z2'], t['eps1'], t['eps2'], supervised=True)
        grads[name] = {k: v.double().numpy() for k, v in orc.last_grads.items() if v is not None}
    model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg, eps=[d['eps1'], d['eps2']])
    grads['product fp32'] = {k: np.asarray(v, np.float64) for k, v in Hh.product_grads(model).items()}
    ref = grads['oracle fp64']
    tot = {n: [0, 0, 0.0, 0.0] for n in ('oracle fp32', 'product fp32')}      # zeros considered, |g| > 1e-7 among them, sum g^2, max
    rows = []
    for k, g64 in ref.items():
        z = np.abs(g64) < 1e-9
        if z.sum() == 0 or k not in grads['product fp32']:
            continue
        r = [k, int(z.sum()), g64.size]
        for n in ('oracle fp32', 'product fp32'):
            g = grads[n][k][z]
            t = tot[n]; t[0] += g.size; t[1] += int((np.abs(g) > 1e-7).sum()); t[2] += float((g * g).sum()); t[3] = max(t[3], float(np.abs(g).max()))
            r += [float(np.sqrt((g * g).mean())), float(np.abs(g).max())]
        rows.append(r)
    rows.sort(key=lambda r: -r[1])
    print('tensor                                   true zeros / elements   oracle fp32 rms, max        product fp32 rms, max')
    for r in rows[:25]:
        print('%-40s %9d / %-9d   %.2e %.2e           %.2e %.2e' % tuple(r))
    for n, t in tot.items():
        print('%s: %d true-zero components, %d of them (%.3f %%) with |g| > 1e-7 (Adam epsilon); rms %.2e, max %.2e'
              % (n, t[0], t[1], 100.0 * t[1] / max(t[0], 1), np.sqrt(t[2] / max(t[0], 1)), t[3]))


if __name__ == '__main__':
    main()
