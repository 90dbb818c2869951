"""Per-volume Dice evaluation on the test volumes (reference model_tester.py:13-85).

On-disk contract: for every modality, every `predict_mask` fusion mode ('simple', 'def', 'max') and both pairings (expert
pairs / pairs randomised within +-2 slices with conf.seed) one folder
    <folder>/test_results_<dataset>_<modality>_<mode>[_rand]/results.csv
holding the header `Vol, Dice, Dice0, ..., Dice{n-1}` and one `%s, %.3f, ...` row per volume.  The PNG dumps of the
reference (plot_images) are visualisation and out of scope."""
import logging
import os

import numpy as np

from . import costs
from .loaders import synthetic

log = logging.getLogger('model_tester')

FUSION_MODES = ('simple', 'def', 'max')


def volume_scores(target, prediction, num_masks):
    """(joint Dice over all organs, [per-organ Dice]) with binarised predictions (costs.py:31-41)"""
    joint = costs.dice(target, prediction, binarise=True)
    per_organ = [costs.dice(target[..., k:k + 1], prediction[..., k:k + 1], binarise=True) for k in range(num_masks)]
    return joint, per_organ


def write_results(path, rows, num_masks):
    cols = ['Vol', 'Dice'] + ['Dice%d' % k for k in range(num_masks)]
    with open(path, 'w') as f:
        f.write(', '.join(cols) + '\n')
        for vol, joint, per_organ in rows:
            f.write(', '.join([str(vol)] + ['%.3f' % v for v in [joint] + list(per_organ)]) + '\n')


class ModelTester(object):
    def __init__(self, model, conf, test_data=None):
        self.model, self.conf = model, conf
        self.test_data = test_data
        self.results = {}

    def make_test_folder(self, modality, suffix=''):
        folder = os.path.join(self.conf.folder, 'test_results_%s_%s_%s' % (self.conf.test_dataset, modality, suffix))
        os.makedirs(folder, exist_ok=True)
        return folder

    def load_test_data(self):
        if self.test_data is None:
            vols = synthetic.splits()['test']
            self.test_data = synthetic.SyntheticPairedData(self.conf.input_shape, self.conf.num_masks, vols,
                                                           self.conf.get('slices_per_volume', 20), 1234 + 202,
                                                           num_modalities=len(self.model.modalities))
        data = self.test_data
        data.crop(self.conf.input_shape[:2])
        return data

    def pairings(self):
        """('' , expert pairs) then ('_rand', a copy with modality-1 slices re-paired inside their volume)"""
        data = self.load_test_data()
        yield '', data
        shuffled = data.copy()
        shuffled.randomise_pairs(length=2, seed=self.conf.seed)
        yield '_rand', shuffled

    def run(self):
        for index, name in enumerate(self.model.modalities):
            log.info('Evaluating model on test data for %s' % name)
            self.test_modality(name, index)
        return self.results

    def test_modality(self, modality, modality_index):
        for tag, data in self.pairings():
            for mode in FUSION_MODES:
                self.test_modality_type(self.make_test_folder(modality, mode + tag), modality_index, mode, data)

    def test_modality_type(self, folder, modality_index, type, test_data):
        if type not in FUSION_MODES + ('maxnostn',):
            raise AssertionError(type)
        rows = []
        for vol in test_data.volumes():
            pair = [test_data.get_volume_images_modi(m, vol) for m in range(len(self.model.modalities))]
            assert pair[0].shape[0] > 0
            truth = test_data.get_volume_masks_modi(modality_index, vol)
            joint, per_organ = volume_scores(truth, self.model.predict_mask(modality_index, type, pair), self.conf.num_masks)
            rows.append((vol, joint, per_organ))
        write_results(os.path.join(folder, 'results.csv'), rows, self.conf.num_masks)
        mean = float(np.mean([r[1] for r in rows]))
        self.results[os.path.basename(folder)] = mean
        log.info('%s - Dice score: %.3f' % (os.path.basename(folder), mean))
