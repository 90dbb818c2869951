import shutil, numpy as np, sys
sys.path.insert(0, '.')
from multimodal_segmentation_amd import nn
from multimodal_segmentation_amd.configuration import dafnet_config_chaos
from tests import helpers as Hh
nn.set_default_device('cuda:0')
from multimodal_segmentation_amd.models.dafnet import DAFNet
from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
conf = Hh.make_conf(dafnet_config_chaos, 64, batch_size=4, epochs=6, slices_per_volume=4, test_dataset='chaos')
conf.folder = '/tmp/mmseg_learn_probe'
shutil.rmtree(conf.folder, ignore_errors=True)
model = DAFNet(conf); model.build()
ex = DAFNetExecutor(conf, model)
import time; t=time.time()
total = ex.train()
print('time', time.time()-t)
for k in ('supervised_Mask','rec_X','val_loss','val_loss_mod1','val_loss_mod2','val_loss_mod2_fused','dis_M','adv_M'):
    print(k, ['%.4f' % v for v in total[k]])
