"""Per-volume Dice evaluation (reference model_tester.py:13-85): for every modality and every fusion mode of
`predict_mask` ('simple', 'def', 'max'), with the expert pairs and with randomised pairs, write
<folder>/test_results_<dataset>_<modality>_<type>[_rand]/results.csv with the columns  Vol, Dice, Dice0..Dice{n-1}.
The PNG dumps of the reference (plot_images) are visualisation and out of scope."""
import logging
import os

import numpy as np

from . import costs
from .loaders import synthetic

log = logging.getLogger('model_tester')


class ModelTester(object):
    def __init__(self, model, conf, test_data=None):
        self.model = model
        self.conf = conf
        self.test_data = test_data
        self.results = {}

    def run(self):
        for modi, mod in enumerate(self.model.modalities):
            log.info('Evaluating model on test data for %s' % mod)
            self.test_modality(mod, modi)
        return self.results

    def make_test_folder(self, modality, suffix=''):
        folder = os.path.join(self.conf.folder, 'test_results_%s_%s_%s' % (self.conf.test_dataset, modality, suffix))
        if not os.path.exists(folder):
            os.makedirs(folder)
        return folder

    def _load(self):
        if self.test_data is not None:
            return self.test_data
        sp = synthetic.splits()
        return synthetic.SyntheticPairedData(self.conf.input_shape, self.conf.num_masks, sp['test'],
                                             self.conf.get('slices_per_volume', 20), 1234 + 202)

    def test_modality(self, modality, modality_index):
        test_data = self._load()
        test_data.crop(self.conf.input_shape[:2])
        for type in ['simple', 'def', 'max']:
            folder = self.make_test_folder(modality, suffix=type)
            self.test_modality_type(folder, modality_index, type, test_data)
        rand = test_data.copy()
        rand.randomise_pairs(length=2, seed=self.conf.seed)
        for type in ['simple', 'def', 'max']:
            folder = self.make_test_folder(modality, suffix=type + '_rand')
            self.test_modality_type(folder, modality_index, type, rand)

    def test_modality_type(self, folder, modality_index, type, test_data):
        assert type in ['simple', 'def', 'max', 'maxnostn']
        num_masks = self.conf.num_masks
        im_dice = {}
        with open(os.path.join(folder, 'results.csv'), 'w') as f:
            f.writelines('Vol, Dice, ' + ', '.join(['Dice%d' % mi for mi in range(num_masks)]) + '\n')
            for vol_i in test_data.volumes():
                vol_image_mod1 = test_data.get_volume_images_modi(0, vol_i)
                vol_image_mod2 = test_data.get_volume_images_modi(1, vol_i)
                assert vol_image_mod1.shape[0] > 0
                vol_mask = test_data.get_volume_masks_modi(modality_index, vol_i)
                prd_mask = self.model.predict_mask(modality_index, type, [vol_image_mod1, vol_image_mod2])
                im_dice[vol_i] = costs.dice(vol_mask, prd_mask, binarise=True)
                sep_dice = [costs.dice(vol_mask[..., mi:mi + 1], prd_mask[..., mi:mi + 1], binarise=True)
                            for mi in range(num_masks)]
                s = '%s, %.3f, ' + ', '.join(['%.3f'] * num_masks) + '\n'
                f.writelines(s % ((str(vol_i), im_dice[vol_i]) + tuple(sep_dice)))
        mean = float(np.mean(list(im_dice.values())))
        self.results[(os.path.basename(folder))] = mean
        print('%s - Dice score: %.3f' % (type, mean))
