"""MMSDNet executor (reference model_executors/mmsdnet_executor.py).

One iteration (`train_batch`, mmsdnet_executor.py:238-331):
  train_batch_generators : supervised_trainer.fit (l_mix > 0) and/or unsupervised_trainer.fit (l_mix < 1), each followed
                           by a Z_Regressor.fit on the six `predict`-mode anatomies with freshly sampled z
  train_batch_mask_discriminator : pool of 4B fake masks (m(s1), m(s2), m(s1_def), m(s1_fused)), sample B, D_Mask fit
As in the DAFNet executor everything stays on the device; only the random draws are made on the host.

More than two modalities (build-defined extension, models/mmsdnet.py): the generator targets follow the ordered
(source -> target) pairs of the graph; the Z-regressor sees the M anatomies + (deformed, fused) of every pair; the fake-mask
pool holds m(s_i) of every modality + m(deformed), m(fused) of the pairs (0 -> j), i.e. (3M - 2) B masks (4B for M = 2).
"""
import logging

import numpy as np
import torch

from ..utils.distributions import NormalDistribution
from .. import ops
from .dafnet_executor import DAFNetExecutor, _dev, _first_channels

log = logging.getLogger('mmsdnet_executor')


class MMSDNetExecutor(DAFNetExecutor):
    """Shares data set-up, validation, testing and the epoch loop with the DAFNet executor; the per-iteration schedule
    is MMSDNet's."""

    def get_loss_names(self):
        return ['adv_M', 'rec_X', 'dis_M', 'val_loss', 'val_loss_mod1', 'val_loss_mod2', 'val_loss_mod2_s1def',
                'val_loss_mod2_fused', 'supervised_Mask', 'loss', 'KL', 'rec_Z']

    # The reference's MMSDNet executor has no stochastic weight averaging: it validates the LIVE models and checkpoints the
    # whole supervised trainer through model.save_models() (mmsdnet_executor.py:166-236)
    def init_swa_models(self):
        pass

    def set_swa_model_weights(self):
        pass

    def get_swa_models(self):
        return []

    def save_models(self, postfix=''):
        from ..parallel import dp
        if dp.is_main():
            self.model.save_models()

    def validate(self, epoch_loss):
        """reference mmsdnet_executor.py:205-236 (modalities 0 and 1, live models)"""
        from .. import costs
        v = self.val_data.copy()
        v.crop(self.conf.input_shape[:2])
        x1, x2 = v.get_images_modi(0), v.get_images_modi(1)
        m1, m2 = v.get_masks_modi(0), v.get_masks_modi(1)
        m = self.model
        s1 = m.Encoders_Anatomy[0].predict(x1)
        s2 = m.Encoders_Anatomy[1].predict(x2)
        s1_def, s_fused = m.Anatomy_Fuser.predict([s1, s2])
        seg = m.Segmentor.predict
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
        """mmsdnet_executor.py:238-252.  conf.multi_stream (build-defined, default on in the reduced-precision modes, bit-identical results): the LAST Z-regressor
        step of the iteration (it trains the decoder and the modality encoder) and the mask-discriminator phase (it reads the
        anatomy encoders, the fuser and the segmentor and trains D_Mask) touch disjoint weights, so they are queued on two HIP
        streams -- same launches, same host order, same random streams (see DAFNetExecutor._train_discriminators)."""
        ms = bool(self.conf.get('multi_stream', self.conf.get('compute_dtype', 'fp32') != 'fp32')) and self.device.type == 'cuda'
        if not ms:
            self.train_batch_generators(epoch_loss)
            self.train_batch_mask_discriminator(epoch_loss)
            return
        deferred = self.train_batch_generators(epoch_loss, defer_last_zreg=True)
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(2)]
        sA, sB = self._streams
        main = torch.cuda.current_stream(self.device)
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            self._z_regressor_step(*deferred)
        with torch.cuda.stream(sB):
            self.train_batch_mask_discriminator(epoch_loss)
        main.wait_stream(sA); main.wait_stream(sB)

    def _five(self, m):
        """Dice only reads the first num_masks channels of target and prediction (costs.py:62-64); the kernel wants both
        with the prediction's 5 channels, so the background channel is appended (it does not enter the loss)."""
        return self._residual(m[..., 0:self.conf.num_masks])

    def _z_regressor_step(self, x_list, epoch_loss, z_list=None):
        """mmsdnet_executor.py:263-276"""
        m = self.model
        batch_size = x_list[0].shape[0]
        norm = NormalDistribution()
        own = [m.Encoders_Anatomy[i].predict(x) for i, x in enumerate(x_list)]
        s_list = list(own)
        for (i, j) in m.pairs():                       # (0, 1) then (1, 0): s1_def, s1_fused, s2_def, s2_fused
            s_list += m.Anatomy_Fuser.predict([own[i], own[j]])
        if z_list is None:
            z_list = [norm.sample((batch_size, self.conf.num_z)).astype(np.float32) for _ in range(m.n_out())]
        h = m.Z_Regressor.fit(s_list + z_list, z_list)
        epoch_loss['rec_Z'].append(self._loss(h, 'loss'))

    def generator_targets(self, x_list, m_list, supervised):
        """[m1, m2, m2, m2, m1, m1] + ones * 6 + [x1, x2, x2, x2, x1, x1] + zeros * 6 for two modalities
        (mmsdnet_executor.py:254-258; unsupervised [m1, m1, m1], 284-288); m_list: 5-channel masks per modality (None where
        unlabelled)"""
        m, n = self.model, self.model.n_out()
        return [m_list[t] for t in m.seg_target_modalities(supervised)] + [1.0] * n + \
               [x_list[t] for t in m.rec_target_modalities()] + [0.0] * n

    def train_batch_generators(self, epoch_loss, eps=None, z_list=None, defer_last_zreg=False):
        """defer_last_zreg: do not run the last Z-regressor step but return its arguments (train_batch queues it beside the
        discriminator phase)"""
        M = self.model.num_mod
        last_sup = defer_last_zreg and not (self.conf.l_mix < 1)
        last_unsup = defer_last_zreg and self.conf.l_mix < 1
        deferred = None
        if self.conf.l_mix > 0:
            batch = next(self.gen_labelled)                              # x_1 .. x_M, m_1 .. m_M
            x_list = [_dev(x, self.device) for x in batch[:M]]
            m_list = [self._five(_dev(mk, self.device)) for mk in batch[M:]]
            h = self.model.supervised_trainer.fit(x_list, self.generator_targets(x_list, m_list, True), eps=eps)
            self._store(h, epoch_loss)
            if last_sup:
                deferred = (x_list, epoch_loss, z_list)
            else:
                self._z_regressor_step(x_list, epoch_loss, z_list)
        if self.conf.l_mix < 1:
            batch = next(self.gen_unlabelled)                            # x_1 .. x_M, m_1
            x_list = [_dev(x, self.device) for x in batch[:M]]
            m_list = [self._five(_dev(batch[M], self.device))] + [None] * (M - 1)
            h = self.model.unsupervised_trainer.fit(x_list, self.generator_targets(x_list, m_list, False), eps=eps)
            self._store(h, epoch_loss)
            if last_unsup:
                deferred = (x_list, epoch_loss, z_list)
            else:
                self._z_regressor_step(x_list, epoch_loss, z_list)
        return deferred

    def _store(self, h, epoch_loss):
        epoch_loss['supervised_Mask'].append(self._loss(h, 'Segmentor_loss'))
        epoch_loss['adv_M'].append(self._loss(h, 'D_Mask_loss'))
        epoch_loss['rec_X'].append(self._loss(h, 'Decoder_loss'))
        epoch_loss['KL'].append(self._loss(h, 'Enc_Modality_loss'))

    def mask_pool(self, *x_list):
        """mmsdnet_executor.py:318-324 -> 4B fake masks: m(s1), m(s2), m(s1_def), m(s1_fused)  [(3M - 2) B for M modalities]"""
        m, nm = self.model, self.conf.num_masks
        fake_s = [m.Encoders_Anatomy[i].predict(x) for i, x in enumerate(x_list)]
        extra = []
        for j in range(1, len(fake_s)):
            extra += m.Anatomy_Fuser.predict([fake_s[0], fake_s[j]])
        # inference-mode BatchNorm has no batch statistics: the segmentations are one batched call
        return ops.slice_channels(m.Segmentor.predict(ops.cat_batch(fake_s + extra)), 0, nm)

    def train_batch_mask_discriminator(self, epoch_loss):
        nm = self.conf.num_masks
        m = _first_channels(_dev(next(self.discriminator_masks), self.device), nm)
        x_list = [_dev(next(gen), self.device) for gen in self.discriminator_image]
        mn = min([x.shape[0] for x in x_list] + [m.shape[0]])
        x_list, m = [x[:mn] for x in x_list], m[:mn]
        pool = self.mask_pool(*x_list)
        h = self.model.D_Mask_trainer.fit([m, self._sample(pool, mn)], [1.0, 0.0])
        epoch_loss['dis_M'].append(self._loss(h, 'D_Mask_loss'))
