"""Command-line entry point with the reference's interface (experiment.py:100-111):

    python experiment.py --config <module in configuration/> --split <int> [--l_mix f] [--test b]
                         [--test_dataset chaos] [--automatedpairing b] [--randomise b]

The run folder name and the config mutations follow Experiment.get_config (experiment.py:31-72):
`<folder>[_randomise][_automatedpairing]_l<l_mix>_<modality>_split<split>` with dots stripped, n_pairs = 3 under automated
pairing else 1, `<folder>/experiment_configuration.json` written before and after training (74-78, 93-97).
"""
import argparse
import importlib
import json
import logging
import os
import subprocess

import numpy as np

from .parallel import dp
from .utils.config import EasyDict

_PKG = __package__


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser(description='multimodal segmentation experiment')
    ap.add_argument('--config', required=True, help='module name under configuration/')
    ap.add_argument('--split', required=True, help='data split index')
    ap.add_argument('--test', type=bool, help='only evaluate on the test volumes')
    ap.add_argument('--test_dataset', choices=['chaos'], help='override the configured test dataset')
    ap.add_argument('--l_mix', help='fraction of labelled volumes')
    ap.add_argument('--automatedpairing', type=bool, help='learn the pairing weights (n_pairs = 3)')
    ap.add_argument('--randomise', type=bool, help='randomise the multimodal pairs')
    return ap.parse_args(argv)


def _flag(config, args, name):
    return bool(config.get(name, False)) or bool(getattr(args, name, None))


def folder_name(base, randomise, automatedpairing, l_mix, modality, split):
    parts = [base]
    if randomise:
        parts.append('randomise')
    if automatedpairing:
        parts.append('automatedpairing')
    parts += ['l%s' % l_mix, str(modality), 'split%s' % split]
    return '_'.join(parts).replace('.', '')


def git_hash():
    try:
        return subprocess.check_output(['git', 'rev-parse', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return ''


def _jsonable(o):
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    raise TypeError(type(o))


def resolve(subpackage, dotted):
    """'dafnet.DAFNet' under models/ or model_executors/ -> class (experiment.py:115-123)"""
    module, cls = dotted.split('.')
    return getattr(importlib.import_module('%s.%s.%s' % (_PKG, subpackage, module)), cls)


class Experiment(object):
    def __init__(self):
        self.log = None

    # ---- configuration --------------------------------------------------------------------------------------------
    def get_config(self, split, args):
        conf = EasyDict(importlib.import_module('%s.configuration.%s' % (_PKG, args.config)).get())
        conf.split = split
        conf.randomise = _flag(conf, args, 'randomise')
        conf.automatedpairing = _flag(conf, args, 'automatedpairing')
        conf.n_pairs = 3 if conf.automatedpairing else 1
        shown = conf.l_mix                       # the folder carries the string as typed on the command line
        if getattr(args, 'l_mix', None) is not None:
            conf.l_mix, shown = float(args.l_mix), args.l_mix
        conf.folder = folder_name(conf.folder, conf.randomise, conf.automatedpairing, shown, conf.modality, split)
        if getattr(args, 'test_dataset', None):
            conf.test_dataset = args.test_dataset
        conf.githash = git_hash()
        self.save_config(conf)
        return conf


    def save_config(self, conf):
        if not dp.is_main():                      # data parallel: rank 0 owns the run folder
            return
        os.makedirs(conf.folder, exist_ok=True)
        with open(os.path.join(conf.folder, 'experiment_configuration.json'), 'w') as f:
            json.dump(dict(conf.items()), f, default=_jsonable)

    def init_logging(self, conf):
        """<folder>/logfile.log at DEBUG level + the console (experiment.py:21-29).  Handlers are attached explicitly, so the
        log file also appears when the embedding process has configured the root logger already."""
        os.makedirs(conf.folder, exist_ok=True)
        root = logging.getLogger()
        root.setLevel(logging.DEBUG)
        if dp.is_main():
            fh = logging.FileHandler(os.path.join(conf.folder, 'logfile.log'))
            fh.setFormatter(logging.Formatter('%(asctime)s %(message)s'))
            root.addHandler(fh)
            root.addHandler(logging.StreamHandler())
        self.log = root
        root.debug(conf.items())
        root.info('---- Setting up experiment at ' + conf.folder + '----')

    # ---- run ----------------------------------------------------------------------------------------------------
    def get_executor(self, conf, test=False):
        model = resolve('models', conf.model)(conf)
        model.build()
        dp.sync_model(model)                      # data parallel: every replica starts from rank 0's weights
        return resolve('model_executors', conf.executor)(conf, model)

    def run_experiment(self, conf, test):
        executor = self.get_executor(conf, test)
        if not test:
            executor.train()
            self.save_config(conf)               # training adds keys (e.g. unlabelled counts)
        if dp.is_main():                          # replicas are identical after training: one rank evaluates and writes
            executor.test()
        dp.host_barrier()                         # not a GPU collective: the other ranks may wait longer than RCCL's watchdog allows

    def run(self, argv=None):
        args = parse_arguments(argv)
        # one process per GPU under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE): join the RCCL group and
        # shard the slice pairs across ranks (parallel/dp.py); a plain `python experiment.py` run is single-GPU
        dp.init_from_env()
        conf = self.get_config(int(args.split), args)
        self.init_logging(conf)
        self.run_experiment(conf, args.test)

    read_console_parameters = staticmethod(parse_arguments)


if __name__ == '__main__':
    Experiment().run()
