"""DAFNet: model wrapper wiring the components into the compiled trainers (reference models/dafnet.py).

Generator trainers (get_params_expert_pairing, dafnet.py:163-222): inputs [x1, x2, z1_input, z2_input]; 20 outputs
(supervised) / 18 (unsupervised: no m2, no m2_s1_def) with the loss table of build_trainers_expertpairs
(dafnet.py:145-149): Segmentor -> combined Dice + 0.01*BCE (w_sup_M), D_Mask -> mse (w_adv_M), Decoder -> mae
(w_rec_X), D_Image1/2 -> mse (w_adv_X), Enc_Modality -> ypred on the KL output (w_kl), ZReconstruct -> mae (w_rec_Z).
supervised_trainer and unsupervised_trainer share the generator weights but own separate Adam states
(dafnet.py:155,161).  The three discriminators are frozen inside the generator trainers (make_trainable(...,
False), dafnet.py:119-121) and trained by their own trainers (dafnet.py:75-115, mmsdnet.py:62-77).
"""
import logging
import os

from .. import costs, nn, ops
from ..model_components import anatomy_fuser, modality_encoder, segmentor, decoder, balancer
from ..model_components.anatomy_encoder import AnatomyEncoders
from .discriminator import Discriminator
from .mmsdnet import MMSDNet, _MuView, _Frozen
from .trainer import Trainer, OutputSpec

log = logging.getLogger('dafnet')


class DAFNet(MMSDNet):
    def __init__(self, conf):
        super(DAFNet, self).__init__(conf)
        self.D_Image1 = None
        self.D_Image2 = None
        self.Balancer = None
        self.D_Image1_trainer = None
        self.D_Image2_trainer = None

    def build(self):
        self.apply_compute_dtype()
        self.build_mask_discriminator()
        self.build_image_discriminator1()
        self.build_image_discriminator2()
        self.build_generators()
        self.apply_loss_scale()
        # only a MISSING checkpoint means "start from scratch" (load_models returns quietly); an unreadable or mismatched
        # one raises instead of silently training from random weights (the reference swallows every exception here,
        # dafnet.py:47-52)
        self.load_models()

    # ---- checkpoint layout <folder>/models/<component> (dafnet.py:54-73) ----------------------------------------------
    def _checkpoint_items(self):
        return [('D_Mask', self.D_Mask), ('D_Image1', self.D_Image1), ('D_Image2', self.D_Image2),
                ('Enc_Anatomy1', self.Encoders_Anatomy[0]), ('Enc_Anatomy2', self.Encoders_Anatomy[1]),
                ('Enc_Modality', self.Enc_Modality), ('Anatomy_Fuser', self.Anatomy_Fuser), ('Segmentor', self.Segmentor),
                ('Decoder', self.Decoder)]

    def load_models(self):
        model_folder = self.conf.folder + '/models/'
        if not os.path.exists(model_folder + 'D_Mask'):
            return
        log.info('Loading trained models from file')
        for fname, m in self._checkpoint_items():
            m.load_weights(model_folder + fname)
        if self.Balancer is not None and os.path.exists(model_folder + 'Balancer'):
            self.Balancer.load_weights(model_folder + 'Balancer')

    def save_models(self, postfix=''):
        model_folder = self.conf.folder + '/models/'
        if not os.path.exists(model_folder):
            os.makedirs(model_folder)
        for fname, m in self._checkpoint_items():
            m.save_weights(model_folder + fname + postfix)
        if self.Balancer is not None:
            self.Balancer.save_weights(model_folder + 'Balancer' + postfix)

    # ---- discriminators ---------------------------------------------------------------------------------------------
    def _build_image_discriminator(self, name):
        params = self.conf.d_image_params
        params['name'] = name
        D = Discriminator(params)
        D.build()
        log.info('Image Discriminator ' + name)
        return D.model, self._d_trainer(D.model, name + '_trainer', self.conf.d_image_params.lr)

    def build_image_discriminator1(self):
        self.D_Image1, self.D_Image1_trainer = self._build_image_discriminator('D_Image1')

    def build_image_discriminator2(self):
        self.D_Image2, self.D_Image2_trainer = self._build_image_discriminator('D_Image2')

    # ---- generators -----------------------------------------------------------------------------------------------
    def build_generators(self):
        assert self.D_Mask is not None, 'Discriminator has not been built yet'
        self.Encoders_Anatomy = AnatomyEncoders(self.modalities).build(self.conf.anatomy_encoder)
        self.Anatomy_Fuser = anatomy_fuser.build(self.conf)
        self.Enc_Modality = modality_encoder.build(self.conf)
        self.Enc_Modality_mu = _MuView(self.Enc_Modality)
        self.Segmentor = segmentor.build(self.conf)
        self.Decoder = decoder.build(self.conf)
        self.Balancer = balancer.build(self.conf)
        self.build_trainers()

    def build_trainers(self):
        self.build_z_regressor()
        if not self.conf.automatedpairing:
            self.build_trainers_expertpairs()
        else:
            self.build_trainers_automatedpairs()

    def build_z_regressor(self):
        self.Z_Regressor = self._z_regressor(2)       # dafnet.py:336-350

    def _expert_graph(self, supervised):
        """get_params_expert_pairing (dafnet.py:163-222)"""
        nm = self.conf.num_masks

        def graph(ins, training=True, eps=None):
            x1, x2, z1_input, z2_input = ins
            eps = eps or [None, None]
            with _Frozen([self.D_Mask, self.D_Image1, self.D_Image2]):
                # encode.  Tensors with several consumers hand every consumer its own alias (ops.Shared): their gradients are then
                # added by ONE launch in the backward pass instead of pairwise by the autograd engine
                S = ops.Shared
                s1_all = self.Encoders_Anatomy[0](x1, training=training)
                s2_all = self.Encoders_Anatomy[1](x2, training=training)
                s1, s2 = S(s1_all, 7), S(s2_all, 7)
                z1, kl1 = self.Enc_Modality(s1.use(), x1, eps=eps[0])
                z2, kl2 = self.Enc_Modality(s2.use(), x2, eps=eps[1])
                z1s, z2s = S(z1, 2), S(z2, 2)
                # segment (BatchNorm batch statistics per call: these stay separate calls)
                m1 = S(self.Segmentor(s1.use(), training=training), 2)
                m2 = S(self.Segmentor(s2.use(), training=training), 2)
                # deform and fuse: both directions in one batched call (per-sample component)
                sd, _ = self.Anatomy_Fuser(ops.cat_batch([s1.use(), s2.use()]), ops.cat_batch([s2.use(), s1.use()]))
                s1_def_all, s2_def_all = ops.split_batch(sd, 2)
                s1_def, s2_def = S(s1_def_all, 2), S(s2_def_all, 2)
                m2_s1_def = S(self.Segmentor(s1_def.use(), training=training), 2)
                m1_s2_def = S(self.Segmentor(s2_def.use(), training=training), 2)
                # decoder: reconstructions, cross-reconstructions and the Z-regressor's decodings (dafnet.py:336-350) are six
                # independent per-sample calls -> one batch of 6B
                ys = self.Decoder(ops.cat_batch([s1.use(), s2.use(), s2_def.use(), s1_def.use(), s1.use(), s2.use()]),
                                  ops.cat_batch([z1s.use(), z2s.use(), z1s.use(), z2s.use(), z1_input, z2_input]))
                y1, y2, y1_s2_def, y2_s1_def, y1_zin, y2_zin = ops.split_batch(ys, 6)
                y1, y2, y1_s2_def, y2_s1_def = S(y1, 2), S(y2, 2), S(y1_s2_def, 2), S(y2_s1_def, 2)
                # GANs (frozen discriminators, no batch statistics): one call per discriminator
                adv_m1, adv_m2, adv_m1_s2_def, adv_m2_s1_def = ops.split_batch(
                    self.D_Mask(ops.cat_batch([ops.slice_channels(m.use(), 0, nm) for m in (m1, m2, m1_s2_def, m2_s1_def)])), 4)
                adv_y1, adv_y1_s2_def = ops.split_batch(self.D_Image1(ops.cat_batch([y1.use(), y1_s2_def.use()])), 2)
                adv_y2, adv_y2_s1_def = ops.split_batch(self.D_Image2(ops.cat_batch([y2.use(), y2_s1_def.use()])), 2)
                # Z-Regressor: Enc_Modality_mu of the decodings of the sampled z
                z1_rec = self.Enc_Modality(s1.use(), y1_zin, mu_only=True)
                z2_rec = self.Enc_Modality(s2.use(), y2_zin, mu_only=True)
                m1, m2, m1_s2_def, m2_s1_def = m1.use(), m2.use(), m1_s2_def.use(), m2_s1_def.use()
                y1, y2, y1_s2_def, y2_s1_def = y1.use(), y2.use(), y1_s2_def.use(), y2_s1_def.use()
                s1, s2, s1_def, s2_def = s1_all, s2_all, s1_def_all, s2_def_all
            all_outputs = [m1, m2, m1_s2_def, m2_s1_def] if supervised else [m1, m1_s2_def]
            all_outputs += [adv_m1, adv_m2, adv_m1_s2_def, adv_m2_s1_def] + \
                           [y1, y2, y1_s2_def, y2_s1_def] + \
                           [adv_y1, adv_y2, adv_y1_s2_def, adv_y2_s1_def] + \
                           [kl1, kl2, z1_rec, z2_rec]
            self.last_factors = {'s1': s1, 's2': s2, 's1_def': s1_def, 's2_def': s2_def, 'z1': z1, 'z2': z2}
            return all_outputs
        return graph

    def _expert_specs(self, supervised):
        c = self.conf
        seg = costs.make_combined_dice_bce(self.num_masks)
        n_seg = 4 if supervised else 2
        return [OutputSpec('Segmentor', seg, c.w_sup_M) for _ in range(n_seg)] + \
               [OutputSpec('D_Mask', 'mse', c.w_adv_M) for _ in range(4)] + \
               [OutputSpec('Decoder', 'mae', c.w_rec_X) for _ in range(4)] + \
               [OutputSpec(n, 'mse', c.w_adv_X) for n in ('D_Image1', 'D_Image2', 'D_Image1', 'D_Image2')] + \
               [OutputSpec('Enc_Modality', costs.ypred, c.w_kl) for _ in range(2)] + \
               [OutputSpec('ZReconstruct', 'mae', c.w_rec_Z) for _ in range(2)]

    # ---- automated pairing (dafnet.py:224-334,352-361) ---------------------------------------------------------------------
    def calculate_weights(self, inputs):
        """Balancer weights [B, n_pairs] of the candidate anatomies inputs[1:] against inputs[0] (dafnet.py:352-361);
        None for a single candidate."""
        if len(inputs) - 1 == 1:
            return None
        return self.Balancer(*inputs)

    def _automated_graph(self, supervised):
        """get_params_automated_pairing (dafnet.py:248-334).  Inputs x1_lst + x2_lst + [m1, (m2,) z1_input, z2_input]; every
        modality contributes n_pairs candidate slices, the first being the expert pair.  The cross-modal segmentation
        and reconstruction terms are computed per sample INSIDE the graph for every candidate and mixed with the
        Balancer weights (outputs 'SegmentorDef' / 'DecoderDef', loss = costs.ypred)."""
        nm, n = self.conf.num_masks, self.conf.n_pairs
        from ..parallel import dp

        def graph(ins, training=True, eps=None):
            x1_lst, x2_lst = list(ins[:n]), list(ins[n:2 * n])
            rest = list(ins[2 * n:])
            m1_input = rest.pop(0)
            m2_input = rest.pop(0) if supervised else None
            z1_input, z2_input = rest
            x1, x2 = x1_lst[0], x2_lst[0]
            eps = eps or [None, None]
            hook = dp.class_sum_hook()
            seg_loss = lambda t, m: ops.seg_loss_per_sample(t, m, nm, costs.lambda_bce, hook)
            with _Frozen([self.D_Mask, self.D_Image1, self.D_Image2]):
                # encode every candidate
                s1_lst = [self.Encoders_Anatomy[0](x, training=training) for x in x1_lst]
                s2_lst = [self.Encoders_Anatomy[1](x, training=training) for x in x2_lst]
                s1, s2 = s1_lst[0], s2_lst[0]
                z1, kl1 = self.Enc_Modality(s1, x1, eps=eps[0])
                z2, kl2 = self.Enc_Modality(s2, x2, eps=eps[1])
                m1 = self.Segmentor(s1, training=training)
                m2 = self.Segmentor(s2, training=training)
                y1 = self.Decoder(s1, z1)
                y2 = self.Decoder(s2, z2)
                adv_m = lambda m: self.D_Mask(ops.slice_channels(m, 0, nm))
                adv_m1, adv_m2 = adv_m(m1), adv_m(m2)
                adv_y1 = self.D_Image1(y1)
                adv_y2 = self.D_Image2(y2)
                # deform every candidate onto the other modality's expert slice; similarity weights
                s1_def_lst = [self.Anatomy_Fuser(s1_i, s2)[0] for s1_i in s1_lst]
                w1_def = self.calculate_weights([s2] + s1_def_lst)
                s2_def_lst = [self.Anatomy_Fuser(s2_i, s1)[0] for s2_i in s2_lst]
                w2_def = self.calculate_weights([s1] + s2_def_lst)
                # weighted cross-reconstruction
                y2_s1_def_lst = [self.Decoder(s, z2) for s in s1_def_lst]
                y1_s2_def_lst = [self.Decoder(s, z1) for s in s2_def_lst]
                y2_s1_def = ops.row_dot(w1_def, [ops.row_mae(x2, y) for y in y2_s1_def_lst])
                y1_s2_def = ops.row_dot(w2_def, [ops.row_mae(x1, y) for y in y1_s2_def_lst])
                # weighted cross-segmentation
                m1_s2_def_lst = [self.Segmentor(s, training=training) for s in s2_def_lst]
                m1_s2_def = ops.row_dot(w2_def, [seg_loss(m1_input, m) for m in m1_s2_def_lst])
                m2_s1_def_lst = [self.Segmentor(s, training=training) for s in s1_def_lst]
                if supervised:
                    m2_s1_def = ops.row_dot(w1_def, [seg_loss(m2_input, m) for m in m2_s1_def_lst])
                # GANs on the expert pair's deformed results
                adv_m2_s1_def, adv_m1_s2_def = adv_m(m2_s1_def_lst[0]), adv_m(m1_s2_def_lst[0])
                adv_y2_s1_def = self.D_Image2(y2_s1_def_lst[0])
                adv_y1_s2_def = self.D_Image1(y1_s2_def_lst[0])
                z1_rec = self.Enc_Modality(s1, self.Decoder(s1, z1_input), mu_only=True)
                z2_rec = self.Enc_Modality(s2, self.Decoder(s2, z2_input), mu_only=True)
            all_outputs = [m1, m2, m1_s2_def, m2_s1_def] if supervised else [m1, m1_s2_def]
            all_outputs += [adv_m1, adv_m2, adv_m1_s2_def, adv_m2_s1_def] + \
                           [y1, y2, y1_s2_def, y2_s1_def] + \
                           [adv_y1, adv_y2, adv_y1_s2_def, adv_y2_s1_def] + \
                           [kl1, kl2, z1_rec, z2_rec]
            self.last_factors = {'s1': s1, 's2': s2, 's1_lst': s1_lst, 's2_lst': s2_lst, 's1_def_lst': s1_def_lst,
                                 's2_def_lst': s2_def_lst, 'w1_def': w1_def, 'w2_def': w2_def, 'z1': z1, 'z2': z2}
            return all_outputs
        return graph

    def _automated_specs(self, supervised):
        """loss table of build_trainers_automatedpairs (dafnet.py:229-235)"""
        c = self.conf
        seg = costs.make_combined_dice_bce(self.num_masks)
        seg_specs = [OutputSpec('Segmentor', seg, c.w_sup_M)] * (2 if supervised else 1) + \
                    [OutputSpec('SegmentorDef', costs.ypred, c.w_sup_M)] * (2 if supervised else 1)
        return seg_specs + \
            [OutputSpec('D_Mask', 'mse', c.w_adv_M) for _ in range(4)] + \
            [OutputSpec('Decoder', 'mae', c.w_rec_X) for _ in range(2)] + \
            [OutputSpec('DecoderDef', costs.ypred, c.w_rec_X) for _ in range(2)] + \
            [OutputSpec(nme, 'mse', c.w_adv_X) for nme in ('D_Image1', 'D_Image2', 'D_Image1', 'D_Image2')] + \
            [OutputSpec('Enc_Modality', costs.ypred, c.w_kl) for _ in range(2)] + \
            [OutputSpec('ZReconstruct', 'mae', c.w_rec_Z) for _ in range(2)]

    def build_trainers_automatedpairs(self):
        assert self.conf.n_pairs == 3, 'the Balancer is wired for 4 inputs = 1 reference + 3 candidates (balancer.py:17-22)'
        gens = self._generator_models() + [self.Balancer]
        frozen = [self.D_Mask, self.D_Image1, self.D_Image2]
        self.unsupervised_trainer = Trainer('unsupervised_trainer', self._automated_graph(False), self._automated_specs(False),
                                            gens, nn.Adam(self.conf.lr), self.num_masks, regularised=frozen)
        self.supervised_trainer = Trainer('supervised_trainer', self._automated_graph(True), self._automated_specs(True),
                                          gens, nn.Adam(self.conf.lr), self.num_masks, regularised=frozen)

    def build_trainers_expertpairs(self):
        """Two compiled models over the SAME generator weights, each with its own Adam (dafnet.py:140-161)."""
        gens = self._generator_models()
        frozen = [self.D_Mask, self.D_Image1, self.D_Image2]
        self.unsupervised_trainer = Trainer('unsupervised_trainer', self._expert_graph(False), self._expert_specs(False),
                                            gens, nn.Adam(self.conf.lr), self.num_masks, regularised=frozen)
        self.supervised_trainer = Trainer('supervised_trainer', self._expert_graph(True), self._expert_specs(True),
                                          gens, nn.Adam(self.conf.lr), self.num_masks, regularised=frozen)
