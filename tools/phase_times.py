#!/usr/bin/env python
"""GPU time of the phases of one DAFNet iteration (HIP events on the compute stream): generator fit, the two pool builds,
the discriminator fits.  python tools/phase_times.py [f32|bf16]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_segmentation_amd import nn, _native
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.models import trainer as T
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from multimodal_segmentation_amd.utils.config import EasyDict

_native.load()
nn.set_default_device('cuda:0')
cfg = dafnet_config_chaos.get()
H, B = 256, 8
cfg['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['output_shape'] = (H, H, 8)
cfg['d_mask_params']['input_shape'] = (H, H, 4); cfg['d_image_params']['input_shape'] = (H, H, 1)
cfg['batch_size'] = B; cfg['n_pairs'] = 1; cfg['folder'] = '/tmp/mmseg_phase'
cfg['compute_dtype'] = 'bf16' if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else 'fp32'
conf = EasyDict(cfg)
model = DAFNet(conf); model.build()
ex = DAFNetExecutor(conf, model); ex.keep_losses_on_device = True
ex.init_train_data(slices_per_volume=4)

marks = []
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(*a, **k); e.record(); marks.append((label, s, e)); return r
    setattr(obj, name, g)
wrap(model.supervised_trainer, 'fit', 'generator fit (fwd + bwd + adam)')
wrap(ex, 'mask_pools', 'mask pools (predict)')
wrap(ex, 'image_pools', 'image pools (predict)')
wrap(model.D_Mask_trainer, 'fit', 'D_Mask fit')
wrap(model.D_Image1_trainer, 'fit', 'D_Image1 fit')
wrap(model.D_Image2_trainer, 'fit', 'D_Image2 fit')
# inside the generator fit: forward graph vs backward
orig_bw = torch.autograd.backward
def bw(*a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig_bw(*a, **k); e.record(); marks.append(('  autograd.backward (all trainers)', s, e)); return r
torch.autograd.backward = bw
losses = {n: [] for n in ex.get_loss_names()}
for _ in range(3):
    ex.train_batch(losses)
torch.cuda.synchronize(); marks.clear()
N = 5
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(N):
    ex.train_batch(losses)
e0.record(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for label, s, e in marks:
    agg[label] = agg.get(label, 0.0) + s.elapsed_time(e)
print('iteration: %.1f ms' % (s0.elapsed_time(e0) / N))
for k, v in agg.items():
    print('%-40s %7.1f ms/iter' % (k, v / N))
