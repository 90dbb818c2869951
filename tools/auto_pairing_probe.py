import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from multimodal_segmentation_amd import nn, _native
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
from multimodal_segmentation_amd.utils.config import EasyDict
_native.load(); nn.set_default_device('cuda:0')
cfg = dafnet_config_chaos.get(); H, B = 256, 8
cfg['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['input_shape'] = (H, H, 1); cfg['anatomy_encoder']['output_shape'] = (H, H, 8)
cfg['d_mask_params']['input_shape'] = (H, H, 4); cfg['d_image_params']['input_shape'] = (H, H, 1)
cfg['batch_size'] = B; cfg['n_pairs'] = 3; cfg['automatedpairing'] = True; cfg['l_mix'] = 0.5; cfg['folder'] = '/tmp/mmseg_auto256'
conf = EasyDict(cfg); model = DAFNet(conf); model.build()
ex = DAFNetExecutor(conf, model); ex.init_train_data(slices_per_volume=8)
losses = {n: [] for n in ex.get_loss_names()}
for i in range(4):
    t = time.time(); ex.train_batch(losses); torch.cuda.synchronize()
    print(i, '%.0f ms' % (1e3 * (time.time() - t)), {k: round(float(v[-1]), 4) for k, v in losses.items() if v}, flush=True)
print('max mem %.1f GB' % (torch.cuda.max_memory_allocated() / 1e9))
assert all(np.isfinite(float(x)) for v in losses.values() for x in v)
