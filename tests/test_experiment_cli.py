"""Entry point contract (reference experiment.py:31-111): argument names, config mutations, run-folder naming, the JSON dump;
and -- on the GPU -- one tiny experiment end to end through the same code path the CLI takes."""
import json
import os
import shutil

import numpy as np
import pytest

from multimodal_segmentation_amd.experiment import Experiment, folder_name, parse_arguments


def test_argument_names_and_folder_naming(tmp_path, monkeypatch):
    args = parse_arguments(['--config', 'dafnet_config_chaos', '--split', '2', '--l_mix', '0.5', '--automatedpairing', '1'])
    assert args.config == 'dafnet_config_chaos' and args.split == '2' and args.l_mix == '0.5' and args.automatedpairing is True
    assert args.test is None and args.randomise is None and args.test_dataset is None
    with pytest.raises(SystemExit):
        parse_arguments(['--split', '0'])                      # --config is required (experiment.py:102)
    assert folder_name('dafnet_chaos', False, False, 1, ['t1', 't2'], 0) == "dafnet_chaos_l1_['t1', 't2']_split0"
    assert folder_name('dafnet_chaos', True, False, '0.5', ['t1', 't2'], 3) == "dafnet_chaos_randomise_l05_['t1', 't2']_split3"
    monkeypatch.chdir(tmp_path)
    conf = Experiment().get_config(2, args)
    assert conf.split == 2 and conf.l_mix == 0.5 and conf.automatedpairing is True and conf.n_pairs == 3
    assert conf.folder == "dafnet_chaos_automatedpairing_l05_['t1', 't2']_split2"
    assert conf.model == 'dafnet.DAFNet' and conf.executor == 'dafnet_executor.DAFNetExecutor'
    dumped = json.load(open(os.path.join(conf.folder, 'experiment_configuration.json')))
    assert dumped['n_pairs'] == 3 and dumped['anatomy_encoder']['out_channels'] == 8 and dumped['input_shape'] == [192, 192, 1]
    # defaults: expert pairing
    conf = Experiment().get_config(0, parse_arguments(['--config', 'mmsdnet_config_chaos', '--split', '0']))
    assert conf.n_pairs == 1 and conf.w_rec_X == 10 and conf.d_mask_params.filters == 4 and 'd_image_params' not in conf


@pytest.mark.gpu
def test_experiment_runs_end_to_end(tmp_path, monkeypatch):
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import _chaos
    nn.set_default_device('cuda:0')
    monkeypatch.chdir(tmp_path)
    # a tiny stand-in for the CHAOS geometry: 64 x 64 slices, 1 epoch, 2 slices per synthetic volume
    real = _chaos.assemble

    def tiny(*a, **k):
        p = real(*a, **k)
        shp = (64, 64, 1)
        p.update(input_shape=shp, epochs=1, batch_size=4, slices_per_volume=2)
        p['anatomy_encoder'].update(input_shape=shp, output_shape=(64, 64, 8))
        p['d_mask_params']['input_shape'] = (64, 64, 4)
        p['d_image_params']['input_shape'] = shp
        return p
    monkeypatch.setattr(_chaos, 'assemble', tiny)
    Experiment().run(['--config', 'dafnet_config_chaos', '--split', '0'])
    folder = "dafnet_chaos_l1_['t1', 't2']_split0"
    assert os.path.exists(os.path.join(folder, 'training.csv')) and os.path.exists(os.path.join(folder, 'logfile.log'))
    assert os.path.exists(os.path.join(folder, 'models', 'Segmentor'))
    rows = open(os.path.join(folder, 'test_results_chaos_t1_simple', 'results.csv')).read().strip().split('\n')
    assert rows[0] == 'Vol, Dice, Dice0, Dice1, Dice2, Dice3' and len(rows) == 4     # three test volumes
    shutil.rmtree(folder, ignore_errors=True)


_COMPAT_SCRIPT = r'''
import importlib, sys, warnings
warnings.simplefilter('error', ImportWarning)
import multimodal_segmentation_amd.compat as compat
compat.install()
# what the reference's experiment.py:113-124 does with conf.model / conf.executor
from configuration import dafnet_config_chaos
conf = dafnet_config_chaos.get()
mod, cls = conf['model'].split('.')
Model = getattr(importlib.import_module('models.' + mod), cls)
mod, cls = conf['executor'].split('.')
Exe = getattr(importlib.import_module('model_executors.' + mod), cls)
# what models/dafnet.py:11-16 and the executors do
from model_components import anatomy_encoder, anatomy_fuser, balancer, decoder, modality_encoder, segmentor
from models.discriminator import Discriminator
from layers import spade, stn_spline
from loaders import loader_factory
from callbacks import swa
from utils import data_utils, distributions
import costs, model_tester
import multimodal_segmentation_amd.model_components.decoder as real_decoder
import multimodal_segmentation_amd.models.dafnet as real_dafnet
assert decoder is real_decoder and Model is real_dafnet.DAFNet
assert decoder.__spec__.name == 'multimodal_segmentation_amd.model_components.decoder'
try:
    import utils.does_not_exist
    raise SystemExit('a module the package does not have must stay unimportable')
except ModuleNotFoundError:
    pass
# decoder.build(conf) exactly as the reference calls it (the CPU stand-in replaces the kernels: no GPU here)
from tests import cpu_backend as cb, helpers as Hh
cb.install()
from multimodal_segmentation_amd import nn
nn.set_default_device('cpu')
c = Hh.make_conf(dafnet_config_chaos, 32)
dec = decoder.build(c)
import numpy as np
y = dec.predict([np.zeros((1, 32, 32, 8), np.float32), np.zeros((1, 8), np.float32)])
assert y.shape == (1, 32, 32, 1)
compat.uninstall()
assert 'model_components' not in sys.modules
print('COMPAT-OK')
'''


def test_reference_top_level_module_names_resolve():
    """callers import `model_components.decoder`, resolve 'models.' + conf.model etc.  Run in a fresh interpreter
    from the repository root (where `python experiment.py` runs) so that nothing imported by the test session interferes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', _COMPAT_SCRIPT], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=300)
    assert out.returncode == 0 and b'COMPAT-OK' in out.stdout, out.stderr.decode()[-2000:]
