"""List the autograd fan-outs of a trainer's graph: tensors that feed several consumers make the autograd engine add their
gradients with a torch kernel (at::native add).  Runs on the test-only CPU stand-in; prints (node, number of gradient edges)."""
import collections
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import cpu_backend as cb, helpers as Hh
cb.install()
from multimodal_segmentation_amd import nn
nn.set_default_device('cpu')
from multimodal_segmentation_amd.configuration import dafnet_config_chaos, dafnet_spade_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet

which = sys.argv[1] if len(sys.argv) > 1 else 'film'
conf = Hh.make_conf(dafnet_spade_config_chaos if which == 'spade' else dafnet_config_chaos, 64)
model = DAFNet(conf)
model.build()
d = Hh.make_step_data(2, 64, 64, seed=7)
for tname in ('supervised_trainer', 'D_Mask_trainer', 'D_Image1_trainer'):
    t = getattr(model, tname)
    if tname == 'supervised_trainer':
        ins = [d['x1'], d['x2'], d['z1'], d['z2']]
    elif tname == 'D_Mask_trainer':
        ins = [d['dm_m1'], d['dm_m2']]
    else:
        ins = [d['x1'], d['x1']]
    ins = [nn.to_device(x, t.device) for x in ins]
    with torch.enable_grad():
        outs = t.graph_fn(ins)
    indeg = collections.Counter()
    seen, stack = set(), [o.grad_fn for o in outs if o.grad_fn is not None]
    for o in outs:
        if o.grad_fn is not None:
            indeg[(o.grad_fn, 0)] += 0
    while stack:
        f = stack.pop()
        if f in seen:
            continue
        seen.add(f)
        for nf, idx in f.next_functions:
            if nf is None:
                continue
            indeg[(nf, idx)] += 1
            stack.append(nf)
    # outputs of the graph also receive a seed gradient
    for o in outs:
        if o.grad_fn is not None:
            indeg[(o.grad_fn, o.output_nr)] += 1
    fan = [(k, v) for k, v in indeg.items() if v > 1 and type(k[0]).__name__ != 'AccumulateGrad']
    print('%s: %d nodes, %d fan-out tensors, %d additions' % (tname, len(seen), len(fan), sum(v - 1 for _, v in fan)))
    names = collections.Counter((type(k[0]).__name__, k[1], v) for k, v in fan)
    for (n, idx, v), c in sorted(names.items()):
        print('   %-28s output %d  consumers %d   x%d' % (n, idx, v, c))
    if '-v' in sys.argv:
        users = collections.defaultdict(list)
        for f in seen:
            for nf, idx in f.next_functions:
                if nf is not None:
                    users[(nf, idx)].append(type(f).__name__)
        for k, v in fan:
            print('     ', type(k[0]).__name__, k[1], '<-', users[k])
