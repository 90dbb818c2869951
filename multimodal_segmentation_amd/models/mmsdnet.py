"""MMSDNet: model wrapper wiring the components into trainers (reference models/mmsdnet.py).

Supervised trainer (mmsdnet.py:146-192): 24 outputs = 6 segmentations (Dice, w_sup_M), 6 adversarial (mse, w_adv_M),
6 reconstructions (mae, w_rec_X), 6 KL (ypred, w_kl).  z is RE-ENCODED from the deformed / fused anatomies
(mmsdnet.py:168-172).  Z_Regressor (194-208) is a separate compiled model over 6 (s, z) pairs.
"""
import logging
import os

import numpy as np

from .. import costs, nn, ops
from ..model_components import anatomy_encoder, anatomy_fuser, modality_encoder, segmentor, decoder
from ..utils.rng import global_rng
from .basenet import BaseNet
from .discriminator import Discriminator
from .trainer import Trainer, OutputSpec

log = logging.getLogger('mmsdnet')


class MMSDNet(BaseNet):
    def __init__(self, conf):
        super(MMSDNet, self).__init__(conf)
        self.modalities = conf.modality
        self.D_Mask = None
        self.Encoders_Anatomy = None
        self.Enc_Modality = None
        self.Enc_Modality_mu = None
        self.Anatomy_Fuser = None
        self.Segmentor = None
        self.Decoder = None
        self.D_Mask_trainer = None
        self.unsupervised_trainer = None
        self.supervised_trainer = None
        self.Z_Regressor = None
        self.num_masks = conf.num_masks if hasattr(conf, 'num_masks') else self.loader.num_masks
        global_rng(conf.seed if hasattr(conf, 'seed') else 10)

    def apply_compute_dtype(self):
        """conf.compute_dtype (build-defined key, default 'fp32'): 'bf16' / 'fp16' run the MFMA products of the fast-path
        convolutions on operands rounded to that type with fp32 accumulation (BASELINE configs #3 / #5: reduced-
        precision compute, fp32 master weights, fp32 gradient all-reduce).  Process-wide switch of the kernel library."""
        ops.set_conv_precision(self.conf.get('compute_dtype', 'fp32'))

    def apply_loss_scale(self):
        """fp16 compute: static loss scale (conf.loss_scale, default 1024) on every trainer, so that small gradients survive
        the rounding of the data-gradient operands to fp16; bf16 / fp32 have fp32's exponent range and need none."""
        scale = float(self.conf.get('loss_scale', 1024.0)) if self.conf.get('compute_dtype', 'fp32') == 'fp16' else 1.0
        for name in ('supervised_trainer', 'unsupervised_trainer', 'Z_Regressor', 'D_Mask_trainer', 'D_Image1_trainer',
                     'D_Image2_trainer'):
            t = getattr(self, name, None)
            if t is not None:
                t.loss_scale = scale

    def build(self):
        self.apply_compute_dtype()
        self.build_mask_discriminator()
        self.build_generators()
        self.apply_loss_scale()
        self.load_models()

    # ---- checkpoint: one file for the whole supervised trainer (mmsdnet.py:42-60) --------------------------------
    def _all_component_models(self):
        return [self.Encoders_Anatomy[0], self.Encoders_Anatomy[1], self.Enc_Modality, self.Anatomy_Fuser, self.Segmentor,
                self.Decoder, self.D_Mask]

    def load_models(self):
        path = self.conf.folder + '/supervised_trainer'
        if os.path.exists(path):
            log.info('Loading trained models from file')
            with np.load(path) as z:
                ws = [z[k] for k in sorted(z.files)]
            i = 0
            for m in self._all_component_models():
                n = len(m.all_params())
                m.set_weights(ws[i:i + n])
                i += n

    def save_models(self):
        log.debug('Saving trained models')
        ws = []
        for m in self._all_component_models():
            ws += m.get_weights()
        with open(self.conf.folder + '/supervised_trainer', 'wb') as f:
            np.savez(f, **{('%05d' % i): w for i, w in enumerate(ws)})

    # ---- discriminator ----------------------------------------------------------------------------------------------
    def _d_trainer(self, D, name, lr):
        """Model([real, fake], [D(real), D(fake)]) compiled with Adam / 'mse' (mmsdnet.py:70-77)"""
        def graph(ins, training=True):
            real, fake = ins                      # no batch statistics in D: real and fake go through as one batch
            if real.shape[0] == fake.shape[0]:
                return ops.split_batch(D(ops.cat_batch([real, fake]), training=training), 2)
            return [D(real, training=training), D(fake, training=training)]
        specs = [OutputSpec(D.name, 'mse', 1.0), OutputSpec(D.name, 'mse', 1.0)]
        return Trainer(name, graph, specs, [D], nn.Adam(lr), regularised=[D])

    def build_mask_discriminator(self):
        D = Discriminator(self.conf.d_mask_params)
        D.build()
        log.info('Mask Discriminator D_M')
        self.D_Mask = D.model
        self.D_Mask_trainer = self._d_trainer(self.D_Mask, 'D_Mask_trainer', self.conf.d_mask_params.lr)

    # ---- generators -----------------------------------------------------------------------------------------------
    def build_generators(self):
        assert self.D_Mask is not None, 'Discriminator has not been built yet'
        self.Encoders_Anatomy = [anatomy_encoder.build(self.conf.anatomy_encoder, 'Enc_Anatomy_%s' % mod)
                                 for mod in self.modalities]
        self.Anatomy_Fuser = anatomy_fuser.build(self.conf)
        self.Enc_Modality = modality_encoder.build(self.conf)
        self.Enc_Modality_mu = _MuView(self.Enc_Modality)
        self.Segmentor = segmentor.build(self.conf)
        self.Decoder = decoder.build(self.conf)
        self.build_unsupervised_trainer()
        self.build_supervised_trainer()
        self.build_z_regressor()

    def _generator_models(self):
        ms = []
        for e in self.Encoders_Anatomy:
            for m in e.owned_models():
                if m not in ms:
                    ms.append(m)
        return ms + [self.Anatomy_Fuser, self.Enc_Modality, self.Segmentor, self.Decoder]

    def _frozen(self, *models):
        return _Frozen(models)

    def _graph(self, supervised):
        nm = self.num_masks

        def graph(ins, training=True, eps=None):
            x_list = ins
            eps = eps or [None] * 6
            with self._frozen(self.D_Mask):
                s_list = [self.Encoders_Anatomy[i](x_list[i], training=training) for i in range(2)]
                z_list = [self.Enc_Modality(s_list[i], x_list[i], eps=eps[i]) for i in range(2)]
                m1, m2 = [self.Segmentor(s, training=training) for s in s_list]
                s1_def, s1_fused = self.Anatomy_Fuser(s_list[0], s_list[1])
                s2_def, s2_fused = self.Anatomy_Fuser(s_list[1], s_list[0])
                fused_seg = [self.Segmentor(s, training=training) for s in [s1_def, s1_fused, s2_def, s2_fused]]
                if supervised:
                    m_list = [m1, m2] + fused_seg
                else:
                    m_list = [m1] + fused_seg[2:]               # masks only for modality 1 (mmsdnet.py:107,116)
                # the frozen discriminator and the decoder have no batch statistics: six calls each -> one batched call
                adv_m_list = ops.split_batch(self.D_Mask(ops.cat_batch([ops.slice_channels(m, 0, nm) for m in [m1, m2] + fused_seg])), 6)
                z_s1def = [self.Enc_Modality(s, x_list[1], eps=eps[2 + i]) for i, s in enumerate([s1_def, s1_fused])]
                z_s2def = [self.Enc_Modality(s, x_list[0], eps=eps[4 + i]) for i, s in enumerate([s2_def, s2_fused])]
                rec_x_list = ops.split_batch(self.Decoder(
                    ops.cat_batch(s_list + [s1_def, s1_fused, s2_def, s2_fused]),
                    ops.cat_batch([z_list[0][0], z_list[1][0], z_s1def[0][0], z_s1def[1][0], z_s2def[0][0], z_s2def[1][0]])), 6)
                diverg = [z_list[i][1] for i in range(2)] + [z_s1def[i][1] for i in range(2)] + [z_s2def[i][1] for i in range(2)]
            return m_list + adv_m_list + rec_x_list + diverg
        return graph

    def _specs(self, supervised):
        c = self.conf
        n_seg = 6 if supervised else 3
        return [OutputSpec('Segmentor', costs.make_dice_loss_fnc(self.num_masks), c.w_sup_M) for _ in range(n_seg)] + \
               [OutputSpec('D_Mask', 'mse', c.w_adv_M) for _ in range(6)] + \
               [OutputSpec('Decoder', 'mae', c.w_rec_X) for _ in range(6)] + \
               [OutputSpec('Enc_Modality', costs.ypred, c.w_kl) for _ in range(6)]

    def build_unsupervised_trainer(self):
        self.unsupervised_trainer = Trainer('unsupervised_trainer', self._graph(False), self._specs(False),
                                            self._generator_models(), nn.Adam(self.conf.lr), self.num_masks,
                                            regularised=[self.D_Mask])

    def build_supervised_trainer(self):
        self.supervised_trainer = Trainer('supervised_trainer', self._graph(True), self._specs(True),
                                          self._generator_models(), nn.Adam(self.conf.lr), self.num_masks,
                                          regularised=[self.D_Mask])

    def _z_regressor(self, num_inputs):
        def graph(ins, training=True):
            s_list, z_list = ins[:num_inputs], ins[num_inputs:]
            xs = [self.Decoder(s, z) for s, z in zip(s_list, z_list)]
            return [self.Enc_Modality(s, x, mu_only=True) for s, x in zip(s_list, xs)]
        specs = [OutputSpec('Enc_Modality_mu', 'mae', self.conf.w_rec_Z) for _ in range(num_inputs)]
        return Trainer('ZReconstruct', graph, specs, [self.Decoder, self.Enc_Modality], nn.Adam(self.conf.lr))

    def build_z_regressor(self):
        self.Z_Regressor = self._z_regressor(len(self.modalities) + 4)      # mmsdnet.py:194-208

    def predict_mask(self, modality_index, type, image_list):
        """reference mmsdnet.py:210-232"""
        assert type in ['simple', 'def', 'max', 'maxnostn']
        idx2 = modality_index
        idx1 = 1 - idx2
        images_mod1 = image_list[idx1]
        images_mod2 = image_list[idx2]
        s1 = self.Encoders_Anatomy[idx1].predict(images_mod1)
        s2 = self.Encoders_Anatomy[idx2].predict(images_mod2)
        if type == 'simple':
            return self.Segmentor.predict(s2)
        elif type == 'def':
            return self.Segmentor.predict(self.Anatomy_Fuser.predict([s1, s2])[0])
        elif type == 'max':
            return self.Segmentor.predict(self.Anatomy_Fuser.predict([s1, s2])[1])
        elif type == 'maxnostn':
            s_max_nostn = np.max([s1, s2], axis=0)
            return self.Segmentor.predict(s_max_nostn)
        raise ValueError(type)


class _MuView(object):
    """Enc_Modality_mu = Model(Enc_Modality.inputs, Enc_Modality.get_layer('z_mean').output) (mmsdnet.py:87)"""

    def __init__(self, enc):
        self.enc = enc
        self.name = 'Enc_Modality_mu'

    def __call__(self, s, x, training=False):
        return self.enc(s, x, training=training, mu_only=True)

    def predict(self, inputs):
        return self.enc.predict(inputs, mu_only=True)


class _Frozen(object):
    """make_trainable(model, False) for the duration of a generator graph (sdnet_utils.py:40-53): the frozen models'
    weights receive no gradient; gradients still flow through them to their inputs."""

    def __init__(self, models):
        self.models = models

    def __enter__(self):
        self.prev = [m.trainable for m in self.models]
        for m in self.models:
            m.trainable = False

    def __exit__(self, *a):
        for m, p in zip(self.models, self.prev):
            m.trainable = p
