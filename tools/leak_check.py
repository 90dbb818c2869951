import os, sys, time
sys.path.insert(0, '.')
import torch
from multimodal_segmentation_amd import nn, _native, ops
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from multimodal_segmentation_amd.utils.config import EasyDict
_native.load(); nn.set_default_device('cuda:0')
cfg = dafnet_config_chaos.get(); H, B = 256, 8
cfg['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['output_shape'] = (H, H, 8)
cfg['d_mask_params']['input_shape'] = (H, H, 4); cfg['d_image_params']['input_shape'] = (H, H, 1)
cfg['batch_size'] = B; cfg['n_pairs'] = 1; cfg['folder'] = '/tmp/mmseg_leak'
for kv in os.environ.get('CONF', '').split(','):       # e.g. CONF=compute_dtype=bf16,act_storage=half,decoder_type=spade,hip_graphs=1
    if '=' in kv:
        k, v = kv.split('=', 1); cfg[k] = (v == '1') if k in ('hip_graphs', 'multi_stream') else v
conf = EasyDict(cfg); model = DAFNet(conf); model.build()
ex = DAFNetExecutor(conf, model); ex.keep_losses_on_device = True; ex.init_train_data(slices_per_volume=4)
losses = {n: [] for n in ex.get_loss_names()}
N_IT = int(os.environ.get('ITERS', 121))
t0 = time.time()
for i in range(N_IT):
    ex.train_batch(losses)
    if i in (5, 20, 60, 120) or i == N_IT - 1:
        torch.cuda.synchronize()
        print(i, 'allocated %.2f GB reserved %.2f GB max %.2f GB; caches: wprep %d bnfold %d ws %d' % (
            torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            len(ops._wprep_cache), len(ops._bnfold_cache), len(ops._workspaces)), flush=True)
    if i == N_IT - 1:
        print('last losses:', {k: round(float(v[-1].item() if hasattr(v[-1], 'item') else v[-1]), 4) for k, v in losses.items() if v}, ' %.1f ms/iter overall' % (1e3 * (time.time() - t0) / N_IT))
    for k in losses: losses[k].clear()
