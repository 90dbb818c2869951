"""MMSDNet: model wrapper wiring the components into trainers (reference models/mmsdnet.py).

Supervised trainer (mmsdnet.py:146-192): 24 outputs = 6 segmentations (Dice, w_sup_M), 6 adversarial (mse, w_adv_M),
6 reconstructions (mae, w_rec_X), 6 KL (ypred, w_kl).  z is RE-ENCODED from the deformed / fused anatomies
(mmsdnet.py:168-172).  Z_Regressor (194-208) is a separate compiled model over 6 (s, z) pairs.

More than two modalities (BASELINE config #5: 3-modality MMSDNet) -- a BUILD-DEFINED EXTENSION, the reference hard-wires
two (mmsdnet.py:105,120-129,160-161).  The two-modality graph is "every modality by itself, then every ORDERED pair
(i -> j): deform anatomy i onto j, fuse, segment both against modality j's masks, re-encode z from modality j's image,
reconstruct modality j" with the pairs (0 -> 1), (1 -> 0).  For M modalities the same recipe runs over all M (M - 1) ordered
pairs in the order (0,1), (0,2), ..., (1,0), (1,2), ...: n_out = M + 2 M (M - 1) outputs of each of the four kinds (6 for
M = 2 -- the reference graph exactly; 15 for M = 3).  Unsupervised: masks exist for modality 0 only, so the Dice terms are
m_0 and the (i -> 0) pairs.  Parity for M > 2 is by construction against the oracle's same extension only.
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
        # conf.act_storage (build-defined, default 'fp32'): 'half' keeps the trunk's activations / gradients in HBM in the 16-bit type
        ops.set_activation_storage(self.conf.get('act_storage', 'fp32') == 'half')
        # the library-wide switch above is the process default; every trainer / component of THIS wrapper also carries the pair and
        # re-enters it on each fit / predict (ops.precision_scope), so a second model of another precision does not disturb this one
        self._precision = (self.conf.get('compute_dtype', 'fp32'), self.conf.get('act_storage', 'fp32') == 'half')
        # conf.sync_bn (build-defined, default False): BatchNorm batch statistics over all data-parallel ranks (parallel/dp.py)
        from ..parallel import dp
        dp.set_sync_bn(self.conf.get('sync_bn', False))

    def apply_loss_scale(self):
        """fp16 compute: static loss scale (conf.loss_scale, default 1024) on every trainer, so that small gradients survive
        the rounding of the data-gradient operands to fp16; bf16 / fp32 have fp32's exponent range and need none."""
        scale = float(self.conf.get('loss_scale', 1024.0)) if self.conf.get('compute_dtype', 'fp32') == 'fp16' else 1.0
        for name in ('supervised_trainer', 'unsupervised_trainer', 'Z_Regressor', 'D_Mask_trainer', 'D_Image1_trainer',
                     'D_Image2_trainer'):
            t = getattr(self, name, None)
            if t is not None:
                t.precision = getattr(self, '_precision', None)
                t.loss_scale = scale
                # conf.hip_graphs (build-defined, default False): record each trainer step into a hipGraph and replay it (graphs.py)
                t.use_graph = bool(self.conf.get('hip_graphs', False))
        for m in self._generator_models() + [d for d in (getattr(self, 'D_Mask', None), getattr(self, 'D_Image1', None),
                                                         getattr(self, 'D_Image2', None), getattr(self, 'Balancer', None)) if d is not None]:
            m.precision = getattr(self, '_precision', None)          # `predict` of a component re-enters this wrapper's precision

    def build(self):
        self.apply_compute_dtype()
        self.build_mask_discriminator()
        self.build_generators()
        self.apply_loss_scale()
        self.load_models()

    # ---- checkpoint: one file for the whole supervised trainer (mmsdnet.py:42-60) --------------------------------
    def _all_component_models(self):
        return list(self.Encoders_Anatomy) + [self.Enc_Modality, self.Anatomy_Fuser, self.Segmentor,
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

    # ---- modality pairs ---------------------------------------------------------------------------------------------
    @property
    def num_mod(self):
        return len(self.modalities)

    def pairs(self):
        """ordered (source -> target) modality pairs in graph order: [(0, 1), (1, 0)] for the reference's two modalities"""
        M = self.num_mod
        return [(i, j) for i in range(M) for j in range(M) if i != j]

    def n_out(self):
        """outputs of each kind: M per-modality + (deformed, fused) per ordered pair (`num_mod * 3` in mmsdnet.py:131-138)"""
        return self.num_mod + 2 * len(self.pairs())

    def seg_target_modalities(self, supervised):
        """modality whose masks are the target of every Dice output, in output order (mmsdnet_executor.py:254,284)"""
        if supervised:
            return list(range(self.num_mod)) + [j for (_, j) in self.pairs() for _k in range(2)]
        return [0] + [0 for (_, j) in self.pairs() if j == 0 for _k in range(2)]

    def rec_target_modalities(self):
        """modality whose image is the target of every reconstruction output (mmsdnet_executor.py:256)"""
        return list(range(self.num_mod)) + [j for (_, j) in self.pairs() for _k in range(2)]

    def _graph(self, supervised):
        nm = self.num_masks

        def graph(ins, training=True, eps=None):
            x_list = ins
            M, pairs, n = self.num_mod, self.pairs(), self.n_out()
            assert len(x_list) == M, '%d inputs for %d modalities' % (len(x_list), M)
            eps = eps or [None] * n
            with self._frozen(self.D_Mask):
                # (tensors with several consumers hand out one alias per consumer, ops.Shared: their gradients are added by one
                # launch in the backward pass instead of pairwise by the autograd engine)
                S = ops.Shared
                s_all = [self.Encoders_Anatomy[i](x_list[i], training=training) for i in range(M)]
                s_list = [S(s, 3 + 2 * (M - 1)) for s in s_all]      # modality encoder, segmentor, decoder, 2 (M - 1) fuser calls
                z_list = [self.Enc_Modality(s_list[i].use(), x_list[i], eps=eps[i]) for i in range(M)]
                m_own = [S(self.Segmentor(s.use(), training=training), 2) for s in s_list]
                # deform + fuse every ordered pair (mmsdnet.py:120-121,160-161 for the two pairs of two modalities)
                fused = []
                for (i, j) in pairs:
                    fused += self.Anatomy_Fuser(s_list[i].use(), s_list[j].use())           # [s_i_def, s_i_fused]
                fused = [S(f, 3) for f in fused]                      # segmentor, modality encoder, decoder
                fused_seg = [S(self.Segmentor(s.use(), training=training), 2) for s in fused]
                # the frozen discriminator and the decoder have no batch statistics: n calls each -> one batched call
                adv_m_list = ops.split_batch(self.D_Mask(ops.cat_batch([ops.slice_channels(m.use(), 0, nm) for m in m_own + fused_seg])), n)
                m_own, fused_seg = [m.use() for m in m_own], [m.use() for m in fused_seg]
                if supervised:
                    m_list = m_own + fused_seg
                else:        # masks only for modality 0 (mmsdnet.py:107,116): m_0 and the (i -> 0) pairs
                    m_list = [m_own[0]] + [fused_seg[2 * p + k] for p, (_, j) in enumerate(pairs) if j == 0 for k in range(2)]
                # z re-encoded from the deformed / fused anatomy and the TARGET modality's image (mmsdnet.py:127-131)
                z_pair = [self.Enc_Modality(s.use(), x_list[pairs[q // 2][1]], eps=eps[M + q]) for q, s in enumerate(fused)]
                rec_x_list = ops.split_batch(self.Decoder(ops.cat_batch([s.use() for s in s_list + fused]),
                                                          ops.cat_batch([z[0] for z in z_list + z_pair])), n)
                diverg = [z[1] for z in z_list + z_pair]
            return m_list + adv_m_list + rec_x_list + diverg
        return graph

    def _specs(self, supervised):
        c = self.conf
        n = self.n_out()
        n_seg = len(self.seg_target_modalities(supervised))
        return [OutputSpec('Segmentor', costs.make_dice_loss_fnc(self.num_masks), c.w_sup_M) for _ in range(n_seg)] + \
               [OutputSpec('D_Mask', 'mse', c.w_adv_M) for _ in range(n)] + \
               [OutputSpec('Decoder', 'mae', c.w_rec_X) for _ in range(n)] + \
               [OutputSpec('Enc_Modality', costs.ypred, c.w_kl) for _ in range(n)]

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
        self.Z_Regressor = self._z_regressor(self.n_out())      # len(modalities) + 4 = 6 in mmsdnet.py:194-208

    def predict_mask(self, modality_index, type, image_list, source_index=None):
        """reference mmsdnet.py:210-232.  `source_index` (more than two modalities only): the modality deformed onto
        `modality_index`; default = the reference's `1 - modality_index` for modalities 0 / 1, modality 0 otherwise."""
        assert type in ['simple', 'def', 'max', 'maxnostn']
        idx2 = modality_index
        idx1 = (1 - idx2 if idx2 in (0, 1) else 0) if source_index is None else source_index
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
