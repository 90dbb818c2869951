"""DAFNet executor: the epoch / batch loop and the per-iteration schedule (reference
model_executors/dafnet_executor.py).

One iteration (`train_batch`, dafnet_executor.py:369-387), expert pairing:
  l_mix > 0 : supervised_trainer.fit  -> D_Mask_trainer.fit x2 -> D_Image1_trainer.fit, D_Image2_trainer.fit
  l_mix < 1 : unsupervised_trainer.fit -> the same three discriminator phases again
The reference performs 20 `predict` + 5 `fit` host<->device round trips per pass; here every tensor stays in HBM:
the fake pools are generated (in `predict` mode, i.e. with the moving BatchNorm statistics), concatenated and
sampled on the device, only the random draws (z, eps, pool indices) are made on the host with numpy exactly like the
reference (utils/distributions.py:9-11, utils/data_utils.py:125-129), and no loss value is read back unless asked.
"""
import logging
import os

import numpy as np
import torch

from .. import costs, nn, ops
from ..callbacks.swa import SWA
from ..loaders import synthetic
from ..model_components import anatomy_encoder, modality_encoder, anatomy_fuser, segmentor, decoder, balancer
from ..model_tester import ModelTester
from ..models.discriminator import Discriminator
from ..parallel import dp
from ..utils import data_utils
from ..utils.distributions import NormalDistribution
from .base_executor import Executor, EarlyStopping

log = logging.getLogger('dafnet_executor')


class DAFNetExecutor(Executor):
    """Train a DAFNet model using parameters stored in the configuration."""

    def __init__(self, conf, model):
        super(DAFNetExecutor, self).__init__(conf, model)
        self.gen_labelled = None          # iterator for labelled data (supervised learning)
        self.gen_unlabelled = None        # iterator for unlabelled data (unsupervised learning)
        self.discriminator_masks = None   # iterator for real masks to train discriminators
        self.discriminator_image = None   # iterators for images to train discriminators
        self.data = None
        self.ul_data = None
        self.device = model.D_Mask.device
        self.keep_losses_on_device = False
        if bool(conf.get('hip_graphs', False)) and self.device.type == 'cuda':
            # conf.hip_graphs: the fake pools (inference passes on a fresh batch) are recorded like the trainer steps (graphs.py)
            from .. import graphs
            self.mask_pools = graphs.GraphedCall(self.mask_pools)
            self.image_pools = graphs.GraphedCall(self.image_pools)
        self.init_swa_models()

    # ---- Stochastic Weight Averaging plumbing (dafnet_executor.py:41-68) -------------------------------------------
    def init_swa_models(self):
        c = self.conf
        self.swa_D_Mask = SWA(40, Discriminator(c.d_mask_params).build, None)
        has_image_d = hasattr(c, 'd_image_params') and getattr(self.model, 'D_Image1', None) is not None
        self.swa_D_Image1 = SWA(40, Discriminator(c.d_image_params).build, None) if has_image_d else None
        self.swa_D_Image2 = SWA(40, Discriminator(c.d_image_params).build, None) if has_image_d else None
        self.swa_Enc_Anatomy1 = SWA(40, anatomy_encoder.build, c.anatomy_encoder)
        self.swa_Enc_Anatomy2 = SWA(40, anatomy_encoder.build, c.anatomy_encoder)
        self.swa_Enc_Modality = SWA(40, modality_encoder.build, c)
        self.swa_Anatomy_Fuser = SWA(40, anatomy_fuser.build, c)
        self.swa_Segmentor = SWA(40, segmentor.build, c)
        self.swa_Decoder = SWA(40, decoder.build, c)
        self.swa_Balancer = SWA(40, balancer.build, c) if getattr(self.model, 'Balancer', None) is not None else None
        self.set_swa_model_weights()

    def set_swa_model_weights(self):
        m = self.model
        pairs = [(self.swa_D_Mask, m.D_Mask), (self.swa_D_Image1, getattr(m, 'D_Image1', None)),
                 (self.swa_D_Image2, getattr(m, 'D_Image2', None)), (self.swa_Enc_Anatomy1, m.Encoders_Anatomy[0]),
                 (self.swa_Enc_Anatomy2, m.Encoders_Anatomy[1]), (self.swa_Enc_Modality, m.Enc_Modality),
                 (self.swa_Anatomy_Fuser, m.Anatomy_Fuser), (self.swa_Segmentor, m.Segmentor), (self.swa_Decoder, m.Decoder),
                 (self.swa_Balancer, getattr(m, 'Balancer', None))]
        for swa, live in pairs:
            if swa is not None:
                swa.model = live

    def _all_models(self):
        m = self.model
        ms = list(m._generator_models())
        for name in ('D_Mask', 'D_Image1', 'D_Image2', 'Balancer'):
            x = getattr(m, name, None)
            if x is not None and x not in ms:
                ms.append(x)
        return ms

    def get_swa_models(self):
        return [s for s in (self.swa_D_Mask, self.swa_D_Image1, self.swa_D_Image2, self.swa_Enc_Anatomy1,
                            self.swa_Enc_Anatomy2, self.swa_Enc_Modality, self.swa_Anatomy_Fuser, self.swa_Segmentor,
                            self.swa_Decoder, self.swa_Balancer) if s is not None]

    # ---- data ----------------------------------------------------------------------------------------------------
    def load_training_volumes(self, slices_per_volume=20, data_seed=1234):
        """All 14 synthetic training volumes (the reference reads CHAOS here: loader.load_all_modalities_concatenated,
        dafnet_executor.py:86); data seed 1234 + rank so that data-parallel ranks hold different slices."""
        from ..parallel import dp
        vols = synthetic.splits()['training']
        return synthetic.SyntheticPairedData(self.conf.input_shape, self.conf.num_masks, vols, slices_per_volume,
                                             data_seed + dp.rank(), num_modalities=self._num_mod())

    def _num_mod(self):
        return len(self.model.modalities)

    def _pairing(self, data, seed=None):
        """randomised / automated pairing of the training pairs (dafnet_executor.py:89-93,130-134)"""
        n_pairs = self.conf.get('n_pairs', 1)
        if self.conf.get('randomise', False):
            data.randomise_pairs(n_pairs - 1, seed=seed)
        elif self.conf.get('automatedpairing', False):
            data.expand_pairs(n_pairs - 1, 0, neighborhood=n_pairs)
            data.expand_pairs(n_pairs - 1, 1, neighborhood=n_pairs)

    def init_train_data(self, device_resident=True, slices_per_volume=20, data_seed=1234):
        """Generators of dafnet_executor.py:70-171 on the synthetic split: round(l_mix * 14) labelled volumes drawn with
        conf.seed, the complement unlabelled; every generator shuffles, batches and rotates on the device
        (utils/augment.py).  `device_resident` is kept for callers of earlier revisions: batches are always produced on
        the executor's device."""
        conf = self.conf
        self.data, self.ul_data, self.data_len = None, None, 0
        if conf.l_mix > 0:                                   # _init_labelled_data_generator (78-99)
            self.data = self.load_training_volumes(slices_per_volume, data_seed)
            self.data.sample(int(np.round(conf.l_mix * self.data.num_volumes)), seed=conf.seed)
            self._pairing(self.data, seed=conf.seed)
            self.data_len = self.data.size()
            d = self.data
            nmod = self._num_mod()
            self.gen_labelled = self.get_data_generator(train_images=[d.get_images_modi(i) for i in range(nmod)],
                                                        train_labels=[d.get_masks_modi(i) for i in range(nmod)])
        if conf.l_mix < 1:                                   # _init_unlabelled_data_generator / _load_unlabelled_data (101-146)
            u = self.load_training_volumes(slices_per_volume, data_seed)
            conf.num_ul_volumes = u.num_volumes
            self._pairing(u)
            if conf.l_mix > 0:
                labelled = u.get_sample_volumes(int(np.round(conf.l_mix * u.num_volumes)), seed=conf.seed)
                u.filter_volumes([v for v in u.volumes() if v not in labelled])
            self.ul_data = u
            conf.unlabelled_image_num = u.size()
            if self.data is None or u.size() > self.data.size():
                self.data_len = u.size()
            self.gen_unlabelled = self.get_data_generator(train_images=[u.get_images_modi(i) for i in range(self._num_mod())],
                                                          train_labels=[u.get_masks_modi(0)])
        self.discriminator_masks = self.get_data_generator(train_labels=[self._load_discriminator_masks()])
        everything = self.load_training_volumes(slices_per_volume, data_seed)    # 'all' data of a modality (141-143,168-171)
        self.discriminator_image = [self.get_data_generator(train_images=[everything.get_images_modi(m)])
                                    for m in range(self._num_mod())]
        self.val_data = synthetic.SyntheticPairedData(conf.input_shape, conf.num_masks, synthetic.splits()['validation'],
                                                      slices_per_volume, data_seed + 101, num_modalities=self._num_mod())
        self.batches = int(np.ceil(self.data_len / float(conf.batch_size)))

    def _load_discriminator_masks(self):
        """real masks for D_Mask: both modalities of the labelled data + modality 1 of the unlabelled (dafnet_executor.py:148-165)"""
        masks = []
        if self.data is not None:
            masks.append(np.concatenate([self.data.get_masks_modi(i) for i in range(self._num_mod())], axis=0))
        if self.ul_data is not None:
            masks.append(self.ul_data.get_masks_modi(0))
        masks = np.concatenate(masks, axis=0)
        assert masks.shape[1:3] == tuple(self.conf.input_shape[:2]), masks.shape
        return masks

    def get_loss_names(self):
        return ['adv_M', 'adv_X1', 'adv_X2', 'rec_X', 'dis_M', 'dis_X1', 'dis_X2',
                'val_loss', 'val_loss_mod1', 'val_loss_mod2',
                'val_loss_mod2_mod1def', 'val_loss_mod1_mod2def', 'val_loss_mod2_fused', 'val_loss_mod1_fused',
                'val_weight_0', 'val_weight_1', 'val_weight_2',
                'supervised_Mask', 'KL', 'rec_Z']

    # ---- epoch loop (dafnet_executor.py:212-284) -------------------------------------------------------------------
    def train(self):
        log.info('Training Model')
        self.init_train_data(slices_per_volume=self.conf.get('slices_per_volume', 20))
        os.makedirs(self.conf.folder, exist_ok=True)
        es = EarlyStopping('val_loss_mod2_fused', min_delta=0.01, patience=60)
        loss_names = self.get_loss_names()
        total_loss = {n: [] for n in loss_names}
        csv_path = self.conf.folder + '/training.csv'
        main = dp.is_main()                    # data parallel: rank 0 writes the csv and the checkpoints
        if main:
            with open(csv_path, 'w') as f:
                f.write('epoch,' + ','.join(loss_names) + '\n')
        for self.epoch in range(self.conf.epochs):
            log.info('Epoch %d/%d' % (self.epoch, self.conf.epochs))
            epoch_loss = {n: [] for n in loss_names}
            for self.batch in range(self.batches):
                self.train_batch(epoch_loss)
            dp.average_state(self._all_models())     # replicas leave the epoch with identical BatchNorm moving statistics
            self.set_swa_model_weights()
            for swa_m in self.get_swa_models():
                swa_m.on_epoch_end(self.epoch)
            self.validate(epoch_loss)
            for n in loss_names:
                total_loss[n].append(np.mean([_f(v) for v in epoch_loss[n]]) if epoch_loss[n] else float('nan'))
            logs = {l: total_loss[l][-1] for l in loss_names}
            log.info(str('Epoch %d/%d: ' + ', '.join([l + ' Loss = %.5f' for l in loss_names])) %
                     ((self.epoch, self.conf.epochs) + tuple(total_loss[l][-1] for l in loss_names)))
            if main:
                with open(csv_path, 'a') as f:
                    f.write('%d,' % self.epoch + ','.join('%.6f' % logs[l] for l in loss_names) + '\n')
            self.save_models()
            if self.stop_criterion(es, logs):
                log.info('Finished training from early stopping criterion')
                for swa_m in self.get_swa_models():
                    swa_m.on_train_end()          # final model parameters = SWA averages (dafnet_executor.py:269-281)
                self.save_models()
                break
        return total_loss

    def save_models(self, postfix=''):
        """Checkpoints hold the SWA clones, one file per component under <folder>/models/ (dafnet_executor.py:286-301)"""
        if not dp.is_main():
            return
        model_folder = self.conf.folder + '/models/'
        if not os.path.exists(model_folder):
            os.makedirs(model_folder)
        names = [('D_Mask', self.swa_D_Mask), ('D_Image1', self.swa_D_Image1), ('D_Image2', self.swa_D_Image2),
                 ('Enc_Anatomy1', self.swa_Enc_Anatomy1), ('Enc_Anatomy2', self.swa_Enc_Anatomy2),
                 ('Enc_Modality', self.swa_Enc_Modality), ('Anatomy_Fuser', self.swa_Anatomy_Fuser),
                 ('Segmentor', self.swa_Segmentor), ('Decoder', self.swa_Decoder), ('Balancer', self.swa_Balancer)]
        for fname, swa in names:
            if swa is not None:
                swa.get_clone_model().save_weights(model_folder + fname + postfix)

    def test(self):
        """Evaluate the model on the test split (reference base_executor.py:89-96 -> model_tester.py)"""
        log.info('Evaluating model on test data')
        tester = ModelTester(self.model, self.conf)
        return tester.run()

    def validate(self, epoch_loss):
        """1 - Dice on the validation split for each modality / deformed / fused input (dafnet_executor.py:303-355)"""
        v = self.val_data.copy()
        if self.conf.get('randomise', False):
            v.randomise_pairs(length=self.conf.n_pairs - 1)
        v.crop(self.conf.input_shape[:2])
        x1, x2 = v.get_images_modi(0), v.get_images_modi(1)
        m1, m2 = v.get_masks_modi(0), v.get_masks_modi(1)
        # validation runs on the SWA clones (dafnet_executor.py:319-335)
        s1 = self.swa_Enc_Anatomy1.get_clone_model().predict(x1)
        s2 = self.swa_Enc_Anatomy2.get_clone_model().predict(x2)
        fuser = self.swa_Anatomy_Fuser.get_clone_model()
        s1_def, s1_fused = fuser.predict([s1, s2])
        s2_def, s2_fused = fuser.predict([s2, s1])
        seg = self.swa_Segmentor.get_clone_model().predict
        l_mod1 = 1 - costs.dice(m1, seg(s1), binarise=True)
        l_mod2 = 1 - costs.dice(m2, seg(s2), binarise=True)
        l_mod2_mod1def = 1 - costs.dice(m2, seg(s1_def), binarise=True)
        l_mod1_mod2def = 1 - costs.dice(m1, seg(s2_def), binarise=True)
        l_mod2_fused = 1 - costs.dice(m2, seg(s1_fused), binarise=True)
        l_mod1_fused = 1 - costs.dice(m1, seg(s2_fused), binarise=True)
        for k, val in (('val_loss_mod1', l_mod1), ('val_loss_mod2', l_mod2), ('val_loss_mod2_mod1def', l_mod2_mod1def),
                       ('val_loss_mod1_mod2def', l_mod1_mod2def), ('val_loss_mod2_fused', l_mod2_fused),
                       ('val_loss_mod1_fused', l_mod1_fused)):
            epoch_loss[k].append(val)
        epoch_loss['val_loss'].append(np.mean([l_mod1, l_mod2, l_mod2_mod1def, l_mod2_fused]))   # dafnet_executor.py:354
        if self.conf.get('automatedpairing', False):
            # mean Balancer weight of every candidate, LIVE encoders / Balancer (dafnet_executor.py:356-367)
            v.expand_pairs(self.conf.n_pairs - 1, 0, neighborhood=self.conf.n_pairs)
            images0 = v.get_images_modi(0)
            s1_list = [self.model.Encoders_Anatomy[0].predict(np.ascontiguousarray(images0[..., i:i + 1]))
                       for i in range(images0.shape[-1])]
            s2 = self.model.Encoders_Anatomy[1].predict(x2)
            weights = self.model.Balancer.predict([s2] + s1_list)
            for j in range(weights.shape[-1]):
                epoch_loss['val_weight_%d' % j].append(float(np.mean(weights[..., j])))

    # ---- one iteration (dafnet_executor.py:369-387) ----------------------------------------------------------------
    def train_batch(self, epoch_loss):
        auto = bool(self.conf.get('automatedpairing', False))
        if self.conf.l_mix > 0:
            (self.train_supervised_automated_pairing if auto else self.train_supervised_expert_pairing)(epoch_loss)
            self._train_discriminators(epoch_loss)
        if self.conf.l_mix < 1:
            (self.train_unsupervised_automated_pairing if auto else self.train_unsupervised_expert_pairing)(epoch_loss)
            self._train_discriminators(epoch_loss)

    def _train_discriminators(self, epoch_loss):
        """mask-discriminator phase, then image-discriminator phase (dafnet_executor.py:378-386).  conf.multi_stream (build-defined;
        default: on in the reduced-precision modes, off in fp32; results are bit-identical either way): the two phases only READ the generator and update different discriminators, so they are queued on two
        HIP streams (the second image discriminator on a third): the discriminators' small launches (a 27 x 27 plane is 92 tiles
        for 256 CUs) then run beside the other phase's full-size inference convolutions instead of alone.  The host issues the
        same launches in the same order (same random streams); scratch buffers and cached weight images are per stream (ops._sid).
        Under data parallelism every rank queues the same collectives in the same order, each behind the launches of its own stream."""
        if not (bool(self.conf.get('multi_stream', self.conf.get('compute_dtype', 'fp32') != 'fp32')) and self.device.type == 'cuda'):
            self.train_batch_mask_discriminator(epoch_loss)
            self.train_batch_image_discriminator(epoch_loss)
            return
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(3)]
        sA, sB, sC = self._streams
        main = torch.cuda.current_stream(self.device)
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            self.train_batch_mask_discriminator(epoch_loss)
        with torch.cuda.stream(sB):
            self.train_batch_image_discriminator(epoch_loss, second_stream=sC)
        main.wait_stream(sA); main.wait_stream(sB)

    def _residual(self, m):
        """add_residual on whatever side the masks live (base_executor.py:83-87): background = 1 - union"""
        if isinstance(m, torch.Tensor):
            return _add_residual_device(m)
        return self.add_residual(m).astype(np.float32)

    def prepare_data_to_train(self, x1_pairs, x2_pairs, m1_pairs, m2_pairs):
        """dafnet_executor.py:482-500: split the candidate pairs stacked on the channel axis, add the background channel,
        draw the z samples of the Z-regressor branch."""
        nm, n_pairs = self.conf.num_masks, self.conf.get('n_pairs', 1)

        def split_images(x):
            if x.shape[-1] == 1:
                return [x]
            return [x[..., i:i + 1].contiguous() if isinstance(x, torch.Tensor) else x[..., i:i + 1] for i in range(n_pairs)]
        x1_list, x2_list = split_images(x1_pairs), split_images(x2_pairs)
        m1 = self._residual(_first_channels(m1_pairs, nm))
        m2 = self._residual(_first_channels(m2_pairs, nm)) if m2_pairs is not None else None
        batch_size = x1_list[0].shape[0]
        norm = NormalDistribution()
        z1 = norm.sample((batch_size, self.conf.num_z)).astype(np.float32)
        z2 = norm.sample((batch_size, self.conf.num_z)).astype(np.float32)
        return m1, m2, x1_list[0], x1_list, x2_list[0], x2_list, z1, z2

    def train_supervised_automated_pairing(self, epoch_loss):
        """dafnet_executor.py:436-458: the weighted cross-modal terms are computed inside the graph, their targets are
        placeholders (zeros)"""
        x1_pairs, x2_pairs, m1_pairs, m2_pairs = next(self.gen_labelled)
        m1, m2, x1, x1_list, x2, x2_list, z1, z2 = self.prepare_data_to_train(x1_pairs, x2_pairs, m1_pairs, m2_pairs)
        h = self.model.supervised_trainer.fit(x1_list + x2_list + [m1, m2, z1, z2],
                                              [m1, m2, 0.0, 0.0] +       # supervised cost
                                              [1.0 for _ in range(4)] +  # mask adversarial
                                              [x1, x2, 0.0, 0.0] +       # reconstruction cost
                                              [1.0 for _ in range(4)] +  # image adversarial
                                              [0.0 for _ in range(2)] +  # KL divergence
                                              [z1, z2])
        self.store_training_losses(h, epoch_loss)

    def train_unsupervised_automated_pairing(self, epoch_loss):
        """dafnet_executor.py:460-480"""
        x1_pairs, x2_pairs, m1_pairs = next(self.gen_unlabelled)
        m1, _, x1, x1_list, x2, x2_list, z1, z2 = self.prepare_data_to_train(x1_pairs, x2_pairs, m1_pairs, None)
        h = self.model.unsupervised_trainer.fit(x1_list + x2_list + [m1, z1, z2],
                                                [m1, 0.0] + [1.0 for _ in range(4)] + [x1, x2, 0.0, 0.0] +
                                                [1.0 for _ in range(4)] + [0.0 for _ in range(2)] + [z1, z2])
        self.store_training_losses(h, epoch_loss)

    def train_supervised_expert_pairing(self, epoch_loss):
        x1, x2, m1, m2 = next(self.gen_labelled)
        m1, m2, x1, _, x2, _, z1, z2 = self.prepare_data_to_train(x1, x2, m1, m2)
        h = self.model.supervised_trainer.fit([x1, x2, z1, z2],
                                              [m1, m2, m1, m2] +      # supervised cost
                                              [1.0 for _ in range(4)] +  # mask adversarial (ones)
                                              [x1, x2, x1, x2] +      # reconstruction cost
                                              [1.0 for _ in range(4)] +  # image adversarial (ones)
                                              [0.0 for _ in range(2)] +  # KL divergence (zeros; costs.ypred ignores it)
                                              [z1, z2])
        self.store_training_losses(h, epoch_loss)

    def train_unsupervised_expert_pairing(self, epoch_loss):
        x1, x2, m1 = next(self.gen_unlabelled)
        m1, _, x1, _, x2, _, z1, z2 = self.prepare_data_to_train(x1, x2, m1, None)
        h = self.model.unsupervised_trainer.fit([x1, x2, z1, z2],
                                                [m1, m1] + [1.0 for _ in range(4)] + [x1, x2, x1, x2] +
                                                [1.0 for _ in range(4)] + [0.0 for _ in range(2)] + [z1, z2])
        self.store_training_losses(h, epoch_loss)

    def store_training_losses(self, h, epoch_loss):
        """dafnet_executor.py:502-509 (keras keeps the LAST output's loss under a duplicated name)"""
        hist = h.history
        g = (lambda k: hist._dev[k]) if self.keep_losses_on_device else (lambda k: hist[k][0])
        epoch_loss['supervised_Mask'].append(g('Segmentor_loss'))
        epoch_loss['adv_M'].append(g('D_Mask_loss'))
        epoch_loss['rec_X'].append(g('Decoder_loss'))
        epoch_loss['adv_X1'].append(g('D_Image1_loss'))
        epoch_loss['adv_X2'].append(g('D_Image2_loss'))
        epoch_loss['KL'].append(g('Enc_Modality_loss'))
        epoch_loss['rec_Z'].append(g('ZReconstruct_loss'))

    # ---- fake pools, generated in `predict` mode and kept on the device ---------------------------------------------
    def mask_pools(self, x1, x2):
        """dafnet_executor.py:524-543 -> (pool of 2B fake masks for modality 1, same for modality 2).  Inference mode has
        no batch statistics, so the two fuser directions and the four segmentations run as one batched call each."""
        m, nm = self.model, self.conf.num_masks
        fake_s1 = m.Encoders_Anatomy[0].predict(x1)
        fake_s2 = m.Encoders_Anatomy[1].predict(x2)
        B = fake_s1.shape[0]
        sd = m.Anatomy_Fuser.predict([ops.cat_batch([fake_s2, fake_s1]), ops.cat_batch([fake_s1, fake_s2])])[0]
        s2_def, s1_def = sd[:B], sd[B:]
        masks = ops.slice_channels(m.Segmentor.predict(ops.cat_batch([fake_s1, s2_def, fake_s2, s1_def])), 0, nm)
        return masks[:2 * B], masks[2 * B:]      # [m(s1); m(s2_def)], [m(s2); m(s1_def)]

    def image_pools(self, x1, x2, eps1=None, eps2=None):
        """dafnet_executor.py:555-575 -> (pool of 3B fake images for modality 1, same for modality 2); the six decodings
        are one batched call"""
        m = self.model
        s1 = m.Encoders_Anatomy[0].predict(x1)
        s2 = m.Encoders_Anatomy[1].predict(x2)
        B = s1.shape[0]
        sd = m.Anatomy_Fuser.predict([ops.cat_batch([s1, s2]), ops.cat_batch([s2, s1])])[0]
        s1_def, s2_def = sd[:B], sd[B:]
        z1, _ = m.Enc_Modality.predict([s1, x1], eps=eps1)
        z2, _ = m.Enc_Modality.predict([s2, x2], eps=eps2)
        ys = m.Decoder.predict([ops.cat_batch([s1, s2_def, s1_def, s2, s1_def, s2_def]), ops.cat_batch([z1, z1, z1, z2, z2, z2])])
        return ys[:3 * B], ys[3 * B:]

    def _sample(self, pool, batch_size):
        """utils.data_utils.sample: np.random.choice(len, size, replace=False), gathered on the device"""
        idx = data_utils.sample_indices(pool.shape[0], batch_size)
        return ops.gather_rows(pool, nn.host_to_device(idx, pool.device, np.int64))

    def train_batch_mask_discriminator(self, epoch_loss):
        """dafnet_executor.py:511-545"""
        nm = self.conf.num_masks
        m1 = _first_channels(_dev(next(self.discriminator_masks), self.device), nm)
        m2 = _first_channels(_dev(next(self.discriminator_masks), self.device), nm)
        x1, x2 = [_dev(next(gen), self.device) for gen in self.discriminator_image]
        mn = min(x1.shape[0], x2.shape[0], m1.shape[0], m2.shape[0])
        x1, x2, m1, m2 = x1[:mn], x2[:mn], m1[:mn], m2[:mn]
        pool1, pool2 = self.mask_pools(x1, x2)
        h = self.model.D_Mask_trainer.fit([m1, self._sample(pool1, mn)], [1.0, 0.0])
        epoch_loss['dis_M'].append(self._loss(h, 'loss'))
        h = self.model.D_Mask_trainer.fit([m2, self._sample(pool2, mn)], [1.0, 0.0])
        epoch_loss['dis_M'].append(self._loss(h, 'loss'))

    def train_batch_image_discriminator(self, epoch_loss, second_stream=None):
        """dafnet_executor.py:547-583 (second_stream: the second image discriminator's step on its own stream, see _train_discriminators)"""
        x1, x2 = [_dev(next(gen), self.device) for gen in self.discriminator_image]
        mn = min(x1.shape[0], x2.shape[0])
        x1, x2 = x1[:mn], x2[:mn]
        y1, y2 = self.image_pools(x1, x2)
        y1 = self._sample(y1, mn)
        y2 = self._sample(y2, mn)
        if second_stream is not None:
            cur = torch.cuda.current_stream(self.device)
            second_stream.wait_stream(cur)
            with torch.cuda.stream(second_stream):
                h2 = self.model.D_Image2_trainer.fit([x2, y2], [1.0, 0.0])
            h = self.model.D_Image1_trainer.fit([x1, y1], [1.0, 0.0])
            epoch_loss['dis_X1'].append(self._loss(h, 'loss'))
            epoch_loss['dis_X2'].append(self._loss(h2, 'loss'))
            cur.wait_stream(second_stream)
            return
        h = self.model.D_Image1_trainer.fit([x1, y1], [1.0, 0.0])
        epoch_loss['dis_X1'].append(self._loss(h, 'loss'))
        h = self.model.D_Image2_trainer.fit([x2, y2], [1.0, 0.0])
        epoch_loss['dis_X2'].append(self._loss(h, 'loss'))

    def _loss(self, h, key):
        return h.history._dev[key] if self.keep_losses_on_device else h.history[key][0]


def _dev(x, device):
    return nn.to_device(x, device)


def _first_channels(t, n):
    """t[..., 0:n] as a contiguous array on whatever side t lives (a kernel of the library on the device, never a torch copy)"""
    if t.shape[-1] == n:
        return t
    if isinstance(t, torch.Tensor):
        return ops.slice_channels(t, 0, n)
    return np.ascontiguousarray(t[..., 0:n])


def _add_residual_device(m):
    """background = 1 except where some mask equals 1 exactly (base_executor.py:83-87; after the bilinear rotation the
    masks carry fractional edge values, which therefore count as background) -- pure data preparation on the device"""
    return ops.add_residual(m)


def _f(v):
    return float(v.item()) if hasattr(v, 'item') else float(v)
