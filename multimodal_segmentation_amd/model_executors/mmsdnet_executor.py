"""MMSDNet executor (reference model_executors/mmsdnet_executor.py).

One iteration (`train_batch`, mmsdnet_executor.py:238-331):
  train_batch_generators : supervised_trainer.fit (l_mix > 0) and/or unsupervised_trainer.fit (l_mix < 1), each followed
                           by a Z_Regressor.fit on the six `predict`-mode anatomies with freshly sampled z
  train_batch_mask_discriminator : pool of 4B fake masks (m(s1), m(s2), m(s1_def), m(s1_fused)), sample B, D_Mask fit
As in the DAFNet executor everything stays on the device; only the random draws are made on the host.
"""
import logging

import numpy as np
import torch

from .. import nn
from ..utils import data_utils
from ..utils.distributions import NormalDistribution
from .dafnet_executor import DAFNetExecutor, _dev

log = logging.getLogger('mmsdnet_executor')


class MMSDNetExecutor(DAFNetExecutor):
    """Shares data set-up, validation, testing and the epoch loop with the DAFNet executor; the per-iteration schedule
    is MMSDNet's."""

    def get_loss_names(self):
        return ['adv_M', 'rec_X', 'dis_M', 'val_loss', 'val_loss_mod1', 'val_loss_mod2', 'val_loss_mod2_s1def',
                'val_loss_mod2_fused', 'supervised_Mask', 'loss', 'KL', 'rec_Z']

    def validate(self, epoch_loss):
        """reference mmsdnet_executor.py:205-236"""
        from .. import costs
        v = self.val_data
        x1, x2 = v.get_images_modi(0), v.get_images_modi(1)
        m1, m2 = v.get_masks_modi(0), v.get_masks_modi(1)
        s1 = self.swa_Enc_Anatomy1.get_clone_model().predict(x1)
        s2 = self.swa_Enc_Anatomy2.get_clone_model().predict(x2)
        s1_def, s_fused = self.swa_Anatomy_Fuser.get_clone_model().predict([s1, s2])
        seg = self.swa_Segmentor.get_clone_model().predict
        l_mod1 = 1 - costs.dice(m1, seg(s1), binarise=True)
        l_mod2 = 1 - costs.dice(m2, seg(s2), binarise=True)
        l_mod2_s1def = 1 - costs.dice(m2, seg(s1_def), binarise=True)
        l_mod2_fused = 1 - costs.dice(m2, seg(s_fused), binarise=True)
        epoch_loss['val_loss_mod2'].append(l_mod2)
        epoch_loss['val_loss_mod2_s1def'].append(l_mod2_s1def)
        epoch_loss['val_loss_mod2_fused'].append(l_mod2_fused)
        epoch_loss['val_loss_mod1'].append(l_mod1)
        epoch_loss['val_loss'].append(np.mean([l_mod1, l_mod2, l_mod2_s1def, l_mod2_fused]))

    def train_batch(self, epoch_loss):
        self.train_batch_generators(epoch_loss)
        self.train_batch_mask_discriminator(epoch_loss)

    def _five(self, m):
        """Dice only reads the first num_masks channels of target and prediction (costs.py:62-64); the kernel wants both
        with the prediction's 5 channels, so the background channel is appended (it does not enter the loss)."""
        return self._residual(m[..., 0:self.conf.num_masks])

    def _z_regressor_step(self, x1, x2, epoch_loss, z_list=None):
        """mmsdnet_executor.py:263-276"""
        m = self.model
        batch_size = x1.shape[0]
        norm = NormalDistribution()
        s_list = [m.Encoders_Anatomy[i].predict(x) for i, x in enumerate([x1, x2])]
        s1_def, s1_fused = m.Anatomy_Fuser.predict(s_list)
        s2_def, s2_fused = m.Anatomy_Fuser.predict(list(reversed(s_list)))
        s_list += [s1_def, s1_fused]
        s_list += [s2_def, s2_fused]
        if z_list is None:
            z_list = [norm.sample((batch_size, self.conf.num_z)).astype(np.float32) for _ in range(6)]
        h = m.Z_Regressor.fit(s_list + z_list, z_list)
        epoch_loss['rec_Z'].append(self._loss(h, 'loss'))

    def train_batch_generators(self, epoch_loss, eps=None, z_list=None):
        ones = 1.0
        if self.conf.l_mix > 0:
            x1, x2, m1, m2 = next(self.gen_labelled)
            x1, x2 = _dev(x1, self.device), _dev(x2, self.device)
            m1, m2 = self._five(_dev(m1, self.device)), self._five(_dev(m2, self.device))
            all_outputs = [m1, m2, m2, m2, m1, m1] + [ones] * 6 + [x1, x2, x2, x2, x1, x1] + [0.0] * 6
            h = self.model.supervised_trainer.fit([x1, x2], all_outputs, eps=eps)
            self._store(h, epoch_loss)
            self._z_regressor_step(x1, x2, epoch_loss, z_list)
        if self.conf.l_mix < 1:
            x1, x2, m1 = next(self.gen_unlabelled)
            x1, x2 = _dev(x1, self.device), _dev(x2, self.device)
            m1 = self._five(_dev(m1, self.device))
            all_outputs = [m1, m1, m1] + [ones] * 6 + [x1, x2, x2, x2, x1, x1] + [0.0] * 6
            h = self.model.unsupervised_trainer.fit([x1, x2], all_outputs, eps=eps)
            self._store(h, epoch_loss)
            self._z_regressor_step(x1, x2, epoch_loss, z_list)

    def _store(self, h, epoch_loss):
        epoch_loss['supervised_Mask'].append(self._loss(h, 'Segmentor_loss'))
        epoch_loss['adv_M'].append(self._loss(h, 'D_Mask_loss'))
        epoch_loss['rec_X'].append(self._loss(h, 'Decoder_loss'))
        epoch_loss['KL'].append(self._loss(h, 'Enc_Modality_loss'))

    def mask_pool(self, x1, x2):
        """mmsdnet_executor.py:318-324 -> 4B fake masks"""
        m, nm = self.model, self.conf.num_masks
        fake_s = [m.Encoders_Anatomy[0].predict(x1), m.Encoders_Anatomy[1].predict(x2)]
        s1_def, s1_fused = m.Anatomy_Fuser.predict(fake_s)
        # inference-mode BatchNorm has no batch statistics: the four segmentations are one batched call
        return m.Segmentor.predict(torch.cat(fake_s + [s1_def, s1_fused], 0))[..., 0:nm].contiguous()

    def train_batch_mask_discriminator(self, epoch_loss):
        nm = self.conf.num_masks
        m = _dev(next(self.discriminator_masks), self.device)[..., 0:nm]
        x1, x2 = [_dev(next(gen), self.device) for gen in self.discriminator_image]
        mn = min(x1.shape[0], x2.shape[0], m.shape[0])
        x1, x2, m = x1[:mn], x2[:mn], m[:mn].contiguous()
        pool = self.mask_pool(x1, x2)
        h = self.model.D_Mask_trainer.fit([m, self._sample(pool, mn)], [1.0, 0.0])
        epoch_loss['dis_M'].append(self._loss(h, 'D_Mask_loss'))
