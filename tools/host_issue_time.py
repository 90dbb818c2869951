#!/usr/bin/env python
"""How long does the HOST need to queue one training iteration?  Issues iterations without synchronising and prints the
per-iteration issue time next to the synchronised step time (host-bound when the two are equal)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_segmentation_amd import nn, _native
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from multimodal_segmentation_amd.utils.config import EasyDict

_native.load()
nn.set_default_device('cuda:0')
cfg = dafnet_config_chaos.get()
H = int(os.environ.get('SIZE', 256)); B = int(os.environ.get('BATCH', 8))
cfg['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['output_shape'] = (H, H, 8)
cfg['d_mask_params']['input_shape'] = (H, H, 4); cfg['d_image_params']['input_shape'] = (H, H, 1)
cfg['batch_size'] = B; cfg['n_pairs'] = 1; cfg['folder'] = '/tmp/mmseg_host_issue'
if os.environ.get('GRAPHS', '0') == '1':
    cfg['hip_graphs'] = True      # trainer steps replayed from hipGraphs (graphs.py)
if os.environ.get('DTYPE'):
    cfg['compute_dtype'] = os.environ['DTYPE']; cfg['act_storage'] = os.environ.get('ACT', 'fp32')
conf = EasyDict(cfg)
model = DAFNet(conf); model.build()
ex = DAFNetExecutor(conf, model); ex.keep_losses_on_device = True
ex.init_train_data(slices_per_volume=2)
losses = {n: [] for n in ex.get_loss_names()}
for _ in range(3):
    ex.train_batch(losses)
torch.cuda.synchronize()
t0 = time.perf_counter(); marks = []
for _ in range(6):
    ex.train_batch(losses); marks.append(time.perf_counter() - t0)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('issue times (cumulative ms):', ['%.1f' % (1e3 * m) for m in marks], ' synchronised total %.1f ms = %.1f ms/iter' % (1e3 * tot, 1e3 * tot / 6))
if os.environ.get('PROFILE', '1') != '1':
    sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    ex.train_batch(losses)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
