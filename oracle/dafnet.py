"""Oracle DAFNet training iteration (models/dafnet.py + model_executors/dafnet_executor.py).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every random draw of the
reference (in-graph eps of `sampling`, numpy z samples, pool indices of
utils/data_utils.py::sample) is an explicit argument so that the product and
the oracle can be driven with identical draws.
"""
from collections import OrderedDict

import torch

from . import models as M
from . import ops as O

GEN_PREFIXES = ('EA0/', 'EA1/', 'EAS/', 'EM/', 'SEG/', 'DEC/', 'FUS/')

DEFAULT_CONF = dict(num_masks=4, num_z=8, w_sup_M=10., w_adv_M=1., w_rec_X=1., w_adv_X=1., w_rec_Z=1.,
                    w_kl=0.1, lr=1e-4, d_lr=1e-4, decoder_type='film')  # configuration/dafnet_config_chaos.py:19-33


class DAFNetOracle(object):
    def __init__(self, P, conf=None):
        self.P = P
        self.conf = dict(DEFAULT_CONF)
        if conf:
            self.conf.update(conf)
        c = self.conf
        # one Adam per compiled trainer (dafnet.py:93,114,155,161; mmsdnet.py:76)
        self.adam = {'sup': O.KerasAdam(c['lr']), 'unsup': O.KerasAdam(c['lr']),
                     'DM': O.KerasAdam(c['d_lr']), 'DI1': O.KerasAdam(c['d_lr']), 'DI2': O.KerasAdam(c['d_lr'])}
        self.decoder = M.decoder_film if c['decoder_type'] == 'film' else M.decoder_spade

    # ---- components in `predict` mode (inference BN) -------------------------------------
    def enc(self, x, mod, training=False, upd=None, soft_only=False):
        return M.anatomy_encoder_dafnet(x, self.P, mod, training, upd, soft_only)

    # ---- generator trainer graph: get_params_expert_pairing (dafnet.py:163-222) ----------
    def generator_forward(self, x1, x2, z1_in, z2_in, eps1, eps2, upd, teacher_s=None):
        """Returns an OrderedDict of every trainer output.  `teacher_s`=(s1, s2) replaces the
        rounded anatomy factors (teacher forcing across the Rounding layer)."""
        P, nm = self.P, self.conf['num_masks']
        out = OrderedDict()
        s1 = self.enc(x1, 0, True, upd)
        s2 = self.enc(x2, 1, True, upd)
        out['s1'], out['s2'] = s1, s2
        if teacher_s is not None:
            s1 = s1 + (teacher_s[0] - s1).detach()
            s2 = s2 + (teacher_s[1] - s2).detach()
        z1, kl1 = M.modality_encoder(s1, x1, eps1, P)
        z2, kl2 = M.modality_encoder(s2, x2, eps2, P)
        m1 = M.segmentor(s1, P, True, upd)
        m2 = M.segmentor(s2, P, True, upd)
        y1 = self.decoder(s1, z1, P)
        y2 = self.decoder(s2, z2, P)
        adv_m1 = M.discriminator(m1[..., :nm], P, 'DM/')
        adv_m2 = M.discriminator(m2[..., :nm], P, 'DM/')
        adv_y1 = M.discriminator(y1, P, 'DI1/')
        adv_y2 = M.discriminator(y2, P, 'DI2/')
        s1_def, _ = M.anatomy_fuser(s1, s2, P)
        s2_def, _ = M.anatomy_fuser(s2, s1, P)
        m2_s1_def = M.segmentor(s1_def, P, True, upd)
        m1_s2_def = M.segmentor(s2_def, P, True, upd)
        y2_s1_def = self.decoder(s1_def, z2, P)
        y1_s2_def = self.decoder(s2_def, z1, P)
        adv_m2_s1_def = M.discriminator(m2_s1_def[..., :nm], P, 'DM/')
        adv_m1_s2_def = M.discriminator(m1_s2_def[..., :nm], P, 'DM/')
        adv_y2_s1_def = M.discriminator(y2_s1_def, P, 'DI2/')
        adv_y1_s2_def = M.discriminator(y1_s2_def, P, 'DI1/')
        # Z_Regressor (dafnet.py:336-350): Decoder then Enc_Modality_mu
        z1_rec = M.modality_encoder_mu(s1, self.decoder(s1, z1_in, P), P)[0]
        z2_rec = M.modality_encoder_mu(s2, self.decoder(s2, z2_in, P), P)[0]
        for k, v in (('m1', m1), ('m2', m2), ('m1_s2_def', m1_s2_def), ('m2_s1_def', m2_s1_def),
                     ('adv_m1', adv_m1), ('adv_m2', adv_m2), ('adv_m1_s2_def', adv_m1_s2_def),
                     ('adv_m2_s1_def', adv_m2_s1_def),
                     ('y1', y1), ('y2', y2), ('y1_s2_def', y1_s2_def), ('y2_s1_def', y2_s1_def),
                     ('adv_y1', adv_y1), ('adv_y2', adv_y2), ('adv_y1_s2_def', adv_y1_s2_def),
                     ('adv_y2_s1_def', adv_y2_s1_def),
                     ('kl1', kl1), ('kl2', kl2), ('z1_rec', z1_rec), ('z2_rec', z2_rec),
                     ('s1_def', s1_def), ('s2_def', s2_def), ('z1', z1), ('z2', z2)):
            out[k] = v
        return out

    # ---- automated pairing: get_params_automated_pairing (dafnet.py:248-334) ---------------------------------------------
    def generator_forward_auto(self, x1_lst, x2_lst, m1_t, m2_t, z1_in, z2_in, eps1, eps2, upd, supervised, teacher_s=None):
        """x1_lst / x2_lst: n_pairs candidate slices per modality, the first being the expert pair.  The weighted
        cross-modal terms (outputs 'SegmentorDef', 'DecoderDef') are per-sample tensors [B,1] computed in the graph.
        `teacher_s` = ([s1 candidates], [s2 candidates]) replaces the rounded anatomies."""
        P, nm = self.P, self.conf['num_masks']
        out = OrderedDict()
        s1_lst = [self.enc(x, 0, True, upd) for x in x1_lst]
        s2_lst = [self.enc(x, 1, True, upd) for x in x2_lst]
        out['s1_lst'], out['s2_lst'] = list(s1_lst), list(s2_lst)
        if teacher_s is not None:
            s1_lst = [s + (t - s).detach() for s, t in zip(s1_lst, teacher_s[0])]
            s2_lst = [s + (t - s).detach() for s, t in zip(s2_lst, teacher_s[1])]
        x1, x2, s1, s2 = x1_lst[0], x2_lst[0], s1_lst[0], s2_lst[0]
        z1, kl1 = M.modality_encoder(s1, x1, eps1, P)
        z2, kl2 = M.modality_encoder(s2, x2, eps2, P)
        m1 = M.segmentor(s1, P, True, upd)
        m2 = M.segmentor(s2, P, True, upd)
        y1 = self.decoder(s1, z1, P)
        y2 = self.decoder(s2, z2, P)
        adv_m1 = M.discriminator(m1[..., :nm], P, 'DM/')
        adv_m2 = M.discriminator(m2[..., :nm], P, 'DM/')
        adv_y1 = M.discriminator(y1, P, 'DI1/')
        adv_y2 = M.discriminator(y2, P, 'DI2/')
        s1_def_lst = [M.anatomy_fuser(s, s2, P)[0] for s in s1_lst]
        w1 = M.balancer(s2, *s1_def_lst, P)                       # calculate_weights([s2] + s1_def_lst), dafnet.py:352-361
        s2_def_lst = [M.anatomy_fuser(s, s1, P)[0] for s in s2_lst]
        w2 = M.balancer(s1, *s2_def_lst, P)
        y2_s1_def_lst = [self.decoder(s, z2, P) for s in s1_def_lst]
        y1_s2_def_lst = [self.decoder(s, z1, P) for s in s2_def_lst]
        # keras Multiply([w [B,1], loss]) + Add (dafnet.py:293-297): per-sample weighted sums, [B,1]
        y2_s1_def = sum(w1[:, j:j + 1] * O.mae_single_input(x2, y) for j, y in enumerate(y2_s1_def_lst))
        y1_s2_def = sum(w2[:, j:j + 1] * O.mae_single_input(x1, y) for j, y in enumerate(y1_s2_def_lst))
        seg = lambda t, m: O.combined_dice_bce_perbatch(t, m, nm).unsqueeze(1)   # [B] -> expand_dims by keras' _Merge
        m1_s2_def_lst = [M.segmentor(s, P, True, upd) for s in s2_def_lst]
        m1_s2_def = sum(w2[:, j:j + 1] * seg(m1_t, m) for j, m in enumerate(m1_s2_def_lst))
        m2_s1_def_lst = [M.segmentor(s, P, True, upd) for s in s1_def_lst]
        m2_s1_def = sum(w1[:, j:j + 1] * seg(m2_t, m) for j, m in enumerate(m2_s1_def_lst)) if supervised else None
        adv_m2_s1_def = M.discriminator(m2_s1_def_lst[0][..., :nm], P, 'DM/')
        adv_m1_s2_def = M.discriminator(m1_s2_def_lst[0][..., :nm], P, 'DM/')
        adv_y2_s1_def = M.discriminator(y2_s1_def_lst[0], P, 'DI2/')
        adv_y1_s2_def = M.discriminator(y1_s2_def_lst[0], P, 'DI1/')
        z1_rec = M.modality_encoder_mu(s1, self.decoder(s1, z1_in, P), P)[0]
        z2_rec = M.modality_encoder_mu(s2, self.decoder(s2, z2_in, P), P)[0]
        for k, v in (('m1', m1), ('m2', m2), ('m1_s2_def', m1_s2_def), ('m2_s1_def', m2_s1_def),
                     ('adv_m1', adv_m1), ('adv_m2', adv_m2), ('adv_m1_s2_def', adv_m1_s2_def),
                     ('adv_m2_s1_def', adv_m2_s1_def),
                     ('y1', y1), ('y2', y2), ('y1_s2_def', y1_s2_def), ('y2_s1_def', y2_s1_def),
                     ('adv_y1', adv_y1), ('adv_y2', adv_y2), ('adv_y1_s2_def', adv_y1_s2_def),
                     ('adv_y2_s1_def', adv_y2_s1_def),
                     ('kl1', kl1), ('kl2', kl2), ('z1_rec', z1_rec), ('z2_rec', z2_rec), ('w1', w1), ('w2', w2)):
            out[k] = v
        out['s1_def_lst'], out['s2_def_lst'] = s1_def_lst, s2_def_lst
        return out

    def generator_losses_auto(self, out, x1, x2, m1_t, m2_t, z1_in, z2_in, supervised):
        """Loss table of build_trainers_automatedpairs (dafnet.py:229-235) in output order."""
        c = self.conf
        nm = c['num_masks']
        terms = []
        seg = lambda t, p: O.combined_dice_bce(t, p, nm)
        if supervised:
            terms += [('Segmentor', c['w_sup_M'], seg(m1_t, out['m1'])), ('Segmentor', c['w_sup_M'], seg(m2_t, out['m2'])),
                      ('SegmentorDef', c['w_sup_M'], out['m1_s2_def'].mean()),
                      ('SegmentorDef', c['w_sup_M'], out['m2_s1_def'].mean())]
        else:
            terms += [('Segmentor', c['w_sup_M'], seg(m1_t, out['m1'])), ('SegmentorDef', c['w_sup_M'], out['m1_s2_def'].mean())]
        for k in ('adv_m1', 'adv_m2', 'adv_m1_s2_def', 'adv_m2_s1_def'):
            terms.append(('D_Mask', c['w_adv_M'], O.mse(torch.ones_like(out[k]), out[k])))
        terms += [('Decoder', c['w_rec_X'], O.mae(x1, out['y1'])), ('Decoder', c['w_rec_X'], O.mae(x2, out['y2'])),
                  ('DecoderDef', c['w_rec_X'], out['y1_s2_def'].mean()), ('DecoderDef', c['w_rec_X'], out['y2_s1_def'].mean())]
        for n, k in (('D_Image1', 'adv_y1'), ('D_Image2', 'adv_y2'), ('D_Image1', 'adv_y1_s2_def'),
                     ('D_Image2', 'adv_y2_s1_def')):
            terms.append((n, c['w_adv_X'], O.mse(torch.ones_like(out[k]), out[k])))
        for k in ('kl1', 'kl2'):
            terms.append(('Enc_Modality', c['w_kl'], out[k].mean()))
        for t, k in ((z1_in, 'z1_rec'), (z2_in, 'z2_rec')):
            terms.append(('ZReconstruct', c['w_rec_Z'], O.mae(t, out[k])))
        total = sum(w * v for _, w, v in terms)
        return total, [(n, v) for n, _, v in terms]

    def generator_step_auto(self, x1_lst, x2_lst, m1_t, m2_t, z1_in, z2_in, eps1, eps2, supervised=True, teacher_s=None):
        """supervised_trainer.fit / unsupervised_trainer.fit of the automated-pairing trainers (Balancer trainable)"""
        if 'sup_auto' not in self.adam:
            self.adam['sup_auto'], self.adam['unsup_auto'] = O.KerasAdam(self.conf['lr']), O.KerasAdam(self.conf['lr'])
        names = M.trainable_names(self.P, GEN_PREFIXES + ('BAL/',))
        self._with_grad(names)
        upd = []
        out = self.generator_forward_auto(x1_lst, x2_lst, m1_t, m2_t, z1_in, z2_in, eps1, eps2, upd, supervised, teacher_s)
        total, terms = self.generator_losses_auto(out, x1_lst[0], x2_lst[0], m1_t, m2_t, z1_in, z2_in, supervised)
        with torch.no_grad():
            reg = sum(M.discriminator_reg(self.P, p) for p in ('DM/', 'DI1/', 'DI2/'))
        self._step('sup_auto' if supervised else 'unsup_auto', names, total)
        self._no_grad(names)
        O.apply_bn_updates(self.P, upd)
        hist = OrderedDict()
        hist['loss'] = float(total.detach() + reg)
        for n, v in terms:
            hist[n + '_loss'] = float(v.detach())
        flat = {}
        for k, v in out.items():
            flat[k] = [t.detach() for t in v] if isinstance(v, list) else (v.detach() if v is not None else None)
        self.last_outputs = flat
        self.last_terms = [(n, float(v.detach())) for n, v in terms]
        return hist

    def generator_losses(self, out, x1, x2, m1_t, m2_t, z1_in, z2_in, supervised):
        """Loss table of build_trainers_expertpairs (dafnet.py:145-149) with the targets of
        train_(un)supervised_expert_pairing (dafnet_executor.py:404-435).  Returns
        (total, [(keras_output_name, value)...]) in output order."""
        c = self.conf
        nm = c['num_masks']
        terms = []
        seg = lambda t, p: O.combined_dice_bce(t, p, nm)
        if supervised:
            seg_pairs = [(m1_t, out['m1']), (m2_t, out['m2']), (m1_t, out['m1_s2_def']), (m2_t, out['m2_s1_def'])]
        else:
            seg_pairs = [(m1_t, out['m1']), (m1_t, out['m1_s2_def'])]
        for t, p in seg_pairs:
            terms.append(('Segmentor', c['w_sup_M'], seg(t, p)))
        for k in ('adv_m1', 'adv_m2', 'adv_m1_s2_def', 'adv_m2_s1_def'):
            terms.append(('D_Mask', c['w_adv_M'], O.mse(torch.ones_like(out[k]), out[k])))
        for t, k in ((x1, 'y1'), (x2, 'y2'), (x1, 'y1_s2_def'), (x2, 'y2_s1_def')):
            terms.append(('Decoder', c['w_rec_X'], O.mae(t, out[k])))
        for n, k in (('D_Image1', 'adv_y1'), ('D_Image2', 'adv_y2'), ('D_Image1', 'adv_y1_s2_def'),
                     ('D_Image2', 'adv_y2_s1_def')):
            terms.append((n, c['w_adv_X'], O.mse(torch.ones_like(out[k]), out[k])))
        for k in ('kl1', 'kl2'):
            terms.append(('Enc_Modality', c['w_kl'], out[k].mean()))   # costs.ypred
        for t, k in ((z1_in, 'z1_rec'), (z2_in, 'z2_rec')):
            terms.append(('ZReconstruct', c['w_rec_Z'], O.mae(t, out[k])))
        total = sum(w * v for _, w, v in terms)
        return total, [(n, v) for n, _, v in terms]

    def _step(self, adam_key, names, loss):
        P = self.P
        grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
        self.last_grads = dict(zip(names, grads))
        self.adam[adam_key].step(P, dict(zip(names, grads)))

    def _with_grad(self, names):
        for k in names:
            self.P[k] = self.P[k].detach().requires_grad_(True)

    def _no_grad(self, names):
        for k in names:
            self.P[k] = self.P[k].detach()

    def generator_step(self, x1, x2, m1_t, m2_t, z1_in, z2_in, eps1, eps2, supervised=True, teacher_s=None):
        """supervised_trainer.fit / unsupervised_trainer.fit: one Adam step, BN moving
        averages updated.  Returns history dict ('<Name>_loss' last-wins, as Keras does with
        duplicate output names) plus 'loss' (incl. the frozen discriminators' regularisers)."""
        names = M.trainable_names(self.P, GEN_PREFIXES)
        self._with_grad(names)
        upd = []
        out = self.generator_forward(x1, x2, z1_in, z2_in, eps1, eps2, upd, teacher_s)
        total, terms = self.generator_losses(out, x1, x2, m1_t, m2_t, z1_in, z2_in, supervised)
        with torch.no_grad():
            reg = sum(M.discriminator_reg(self.P, p) for p in ('DM/', 'DI1/', 'DI2/'))
        self._step('sup' if supervised else 'unsup', names, total)
        self._no_grad(names)
        O.apply_bn_updates(self.P, upd)
        hist = OrderedDict()
        hist['loss'] = float(total.detach() + reg)
        for n, v in terms:
            hist[n + '_loss'] = float(v.detach())
        self.last_outputs = {k: v.detach() for k, v in out.items()}
        self.last_terms = [(n, float(v.detach())) for n, v in terms]
        return hist

    # ---- discriminators ------------------------------------------------------------------
    def discriminator_step(self, prefix, real, fake):
        """D_*_trainer.fit([real, fake], [ones, zeros]) (dafnet_executor.py:534,544,578,581)."""
        names = M.trainable_names(self.P, (prefix,))
        self._with_grad(names)
        d_real = M.discriminator(real, self.P, prefix)
        d_fake = M.discriminator(fake, self.P, prefix)
        l_real = O.mse(torch.ones_like(d_real), d_real)
        l_fake = O.mse(torch.zeros_like(d_fake), d_fake)
        reg = M.discriminator_reg(self.P, prefix)
        total = l_real + l_fake + reg
        self._step(prefix[:-1], names, total)
        self._no_grad(names)
        return {'loss': float(total.detach()), 'real_loss': float(l_real.detach()),
                'fake_loss': float(l_fake.detach()), 'reg': float(reg.detach())}

    @torch.no_grad()
    def mask_pools(self, x1, x2):
        """Fake-mask pools of train_batch_mask_discriminator (dafnet_executor.py:524-543);
        every call is a `predict` (inference-mode BN)."""
        P, nm = self.P, self.conf['num_masks']
        s1, s2 = self.enc(x1, 0), self.enc(x2, 1)
        fake_m1 = M.segmentor(s1, P, False, None)
        s2_def, _ = M.anatomy_fuser(s2, s1, P)
        fake_m1_from_s2 = M.segmentor(s2_def, P, False, None)
        pool1 = torch.cat([fake_m1[..., :nm], fake_m1_from_s2[..., :nm]], 0)
        fake_m2 = M.segmentor(s2, P, False, None)
        s1_def, _ = M.anatomy_fuser(s1, s2, P)
        fake_m2_from_s1 = M.segmentor(s1_def, P, False, None)
        pool2 = torch.cat([fake_m2[..., :nm], fake_m2_from_s1[..., :nm]], 0)
        return pool1, pool2

    @torch.no_grad()
    def image_pools(self, x1, x2, eps1, eps2):
        """Fake-image pools of train_batch_image_discriminator (dafnet_executor.py:555-575)."""
        P = self.P
        s1, s2 = self.enc(x1, 0), self.enc(x2, 1)
        s1_def = M.anatomy_fuser(s1, s2, P)[0]
        s2_def = M.anatomy_fuser(s2, s1, P)[0]
        z1, _ = M.modality_encoder(s1, x1, eps1, P)
        z2, _ = M.modality_encoder(s2, x2, eps2, P)
        y1 = torch.cat([self.decoder(s1, z1, P), self.decoder(s2_def, z1, P), self.decoder(s1_def, z1, P)], 0)
        y2 = torch.cat([self.decoder(s2, z2, P), self.decoder(s1_def, z2, P), self.decoder(s2_def, z2, P)], 0)
        return y1, y2

    def train_batch(self, d, supervised=True):
        """One pass of DAFNetExecutor.train_batch's l_mix branch (dafnet_executor.py:369-387):
        generator fit, mask-D x2, image-D x2.  `d` is a dict of torch tensors:
          x1,x2,m1,m2 (m with background channel), z1,z2, eps1,eps2   -- generator step
          dm_m1, dm_m2 (real masks, 4 ch), dm_x1, dm_x2, dm_idx1, dm_idx2   -- mask D
          di_x1, di_x2, di_eps1, di_eps2, di_idx1, di_idx2                  -- image D
        Returns the losses in get_loss_names() vocabulary."""
        h = self.generator_step(d['x1'], d['x2'], d['m1'], d.get('m2'), d['z1'], d['z2'], d['eps1'], d['eps2'],
                                supervised)
        res = OrderedDict()
        res['supervised_Mask'] = h['Segmentor_loss']; res['adv_M'] = h['D_Mask_loss']
        res['rec_X'] = h['Decoder_loss']; res['adv_X1'] = h['D_Image1_loss']; res['adv_X2'] = h['D_Image2_loss']
        res['KL'] = h['Enc_Modality_loss']; res['rec_Z'] = h['ZReconstruct_loss']; res['gen_total'] = h['loss']
        pool1, pool2 = self.mask_pools(d['dm_x1'], d['dm_x2'])
        res['dis_M_1'] = self.discriminator_step('DM/', d['dm_m1'], pool1[d['dm_idx1']])['loss']
        res['dis_M_2'] = self.discriminator_step('DM/', d['dm_m2'], pool2[d['dm_idx2']])['loss']
        y1, y2 = self.image_pools(d['di_x1'], d['di_x2'], d['di_eps1'], d['di_eps2'])
        res['dis_X1'] = self.discriminator_step('DI1/', d['di_x1'], y1[d['di_idx1']])['loss']
        res['dis_X2'] = self.discriminator_step('DI2/', d['di_x2'], y2[d['di_idx2']])['loss']
        return res
