"""Oracle MMSDNet training iteration (models/mmsdnet.py + model_executors/mmsdnet_executor.py).

TEST INFRASTRUCTURE (see oracle/__init__.py).  All random draws are explicit arguments.
Graph (mmsdnet.py:146-192): two SEPARATE full UNets, 6 segmentations, 6 adversarial outputs, 6 reconstructions whose
z is RE-ENCODED from the deformed / fused anatomies (168-172), 6 KL terms; loss table: Dice only (w_sup_M = 10),
mse (w_adv_M = 1), mae (w_rec_X = 10), ypred (w_kl = 0.1) (configuration/mmsdnet_config_chaos.py:19-24).
Z_Regressor (194-208) is a separately compiled model over 6 (s, z) pairs with its own Adam.

M > 2 modalities (BASELINE config #5) is a BUILD-DEFINED extension -- the reference hard-wires two (mmsdnet.py:105,
120-129,160-161): the per-pair block of the reference graph is applied to every ordered pair (i -> j) in the order
(0,1), (0,2), ..., (1,0), ...; for M = 2 this is the reference graph itself.  PARITY UNPINNED by construction for M > 2.
"""
from collections import OrderedDict

import torch

from . import models as M
from . import ops as O

GEN_PREFIXES = ('EA0/', 'EA1/', 'EM/', 'SEG/', 'DEC/', 'FUS/')     # two modalities; see MMSDNetOracle.gen_prefixes
DEFAULT_CONF = dict(num_masks=4, num_z=8, w_sup_M=10., w_adv_M=1., w_rec_X=10., w_rec_Z=1., w_kl=0.1, lr=1e-4, d_lr=1e-4,
                    decoder_type='film')


class MMSDNetOracle(object):
    def __init__(self, P, conf=None):
        self.P = P
        self.conf = dict(DEFAULT_CONF)
        if conf:
            self.conf.update(conf)
        c = self.conf
        self.adam = {'sup': O.KerasAdam(c['lr']), 'unsup': O.KerasAdam(c['lr']), 'zreg': O.KerasAdam(c['lr']),
                     'DM': O.KerasAdam(c['d_lr'])}
        self.decoder = M.decoder_film if c['decoder_type'] == 'film' else M.decoder_spade
        self.num_mod = sum(1 for i in range(8) if ('EA%d/conv_anatomy/kernel' % i) in P)
        self.pairs = [(i, j) for i in range(self.num_mod) for j in range(self.num_mod) if i != j]
        self.gen_prefixes = tuple('EA%d/' % i for i in range(self.num_mod)) + ('EM/', 'SEG/', 'DEC/', 'FUS/')

    def enc(self, x, mod, training=False, upd=None, soft_only=False):
        return M.anatomy_encoder_mmsdnet(x, self.P, mod, training, upd, soft_only)

    def generator_forward(self, x, eps, upd, supervised=True, teacher_s=None):
        """x: list of the M modality images.  eps: list of M + 2 * len(pairs) [B, num_z] draws in Enc_Modality call order
        (s_1 .. s_M, then (deformed, fused) of every ordered pair; two modalities: s1, s2, s1_def, s1_fused, s2_def, s2_fused)."""
        P, nm, nmod = self.P, self.conf['num_masks'], self.num_mod
        s = [self.enc(x[i], i, True, upd) for i in range(nmod)]
        out = OrderedDict(('s%d' % (i + 1), s[i]) for i in range(nmod))
        if teacher_s is not None:
            s = [s[i] + (teacher_s[i] - s[i]).detach() for i in range(nmod)]
        z = [M.modality_encoder(s[i], x[i], eps[i], P) for i in range(nmod)]
        m_own = [M.segmentor(s[i], P, True, upd) for i in range(nmod)]
        adv = lambda m: M.discriminator(m[..., :nm], P, 'DM/')
        rec = [self.decoder(s[i], z[i][0], P) for i in range(nmod)]
        fused = []
        for (i, j) in self.pairs:                              # mmsdnet.py:160-161 for (0, 1) and (1, 0)
            fused += list(M.anatomy_fuser(s[i], s[j], P))
        fseg = [M.segmentor(a, P, True, upd) for a in fused]
        if supervised:
            m_list = m_own + fseg
        else:                                                  # masks of modality 0 only (mmsdnet.py:107,116)
            m_list = [m_own[0]] + [fseg[2 * p + k] for p, (_, j) in enumerate(self.pairs) if j == 0 for k in range(2)]
        adv_list = [adv(m) for m in m_own] + [adv(m) for m in fseg]
        # z re-encoded from the deformed / fused anatomy with the TARGET modality's image (mmsdnet.py:168-172)
        z_pair = [M.modality_encoder(a, x[self.pairs[q // 2][1]], eps[nmod + q], P) for q, a in enumerate(fused)]
        rec += [self.decoder(a, z_pair[q][0], P) for q, a in enumerate(fused)]
        kls = [zz[1] for zz in z + z_pair]
        out.update(m_list=m_list, adv_list=adv_list, rec_list=rec, kl_list=kls, fused=fused)
        return out

    def generator_step(self, x1, x2, m1_t, m2_t, eps, supervised=True, teacher_s=None):
        """two-modality signature kept for the existing callers; see generator_step_n"""
        return self.generator_step_n([x1, x2], [m1_t, m2_t], eps, supervised, teacher_s)

    def generator_step_n(self, x, m_t, eps, supervised=True, teacher_s=None):
        """supervised_trainer.fit([x1, x2], [m1, m2, m2, m2, m1, m1] + ones*6 + [x1, x2, x2, x2, x1, x1] + zeros*6)
        (mmsdnet_executor.py:254-258); unsupervised: [m1, m1, m1] (284-288).  M modalities: per-modality targets, then the
        TARGET modality's masks / image twice for every ordered pair."""
        c, nm, nmod = self.conf, self.conf['num_masks'], self.num_mod
        names = M.trainable_names(self.P, self.gen_prefixes)
        for k in names:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        upd = []
        out = self.generator_forward(x, eps, upd, supervised, teacher_s)
        pair_t = [j for (_, j) in self.pairs for _k in range(2)]
        if supervised:
            seg_t = [m_t[i] for i in range(nmod)] + [m_t[j] for j in pair_t]
        else:
            seg_t = [m_t[0]] * len(out['m_list'])
        rec_t = [x[i] for i in range(nmod)] + [x[j] for j in pair_t]
        terms = [('Segmentor', c['w_sup_M'], O.dice_loss(t, p, nm)) for t, p in zip(seg_t, out['m_list'])]
        terms += [('D_Mask', c['w_adv_M'], O.mse(torch.ones_like(a), a)) for a in out['adv_list']]
        terms += [('Decoder', c['w_rec_X'], O.mae(t, y)) for t, y in zip(rec_t, out['rec_list'])]
        terms += [('Enc_Modality', c['w_kl'], k.mean()) for k in out['kl_list']]
        total = sum(w * v for _, w, v in terms)
        grads = torch.autograd.grad(total, [self.P[k] for k in names], allow_unused=True)
        self.last_grads = dict(zip(names, grads))
        self.adam['sup' if supervised else 'unsup'].step(self.P, self.last_grads)
        for k in names:
            self.P[k] = self.P[k].detach()
        O.apply_bn_updates(self.P, upd)
        with torch.no_grad():
            reg = M.discriminator_reg(self.P, 'DM/')
        hist = OrderedDict(loss=float(total.detach() + reg))
        for n, _, v in terms:
            hist[n + '_loss'] = float(v.detach())
        self.last_outputs = out
        return hist

    @torch.no_grad()
    def zreg_inputs(self, *x):
        """`predict`-mode anatomies fed to Z_Regressor.fit (mmsdnet_executor.py:264-270)."""
        s = [self.enc(xi, i) for i, xi in enumerate(x)]
        out = list(s)
        for (i, j) in self.pairs:
            out += list(M.anatomy_fuser(s[i], s[j], self.P))
        return out

    def zreg_step(self, s_list, z_list):
        """Z_Regressor.fit(s_list + z_list, z_list): Decoder then Enc_Modality_mu, mae, w_rec_Z (mmsdnet.py:194-208)."""
        names = M.trainable_names(self.P, ('DEC/', 'EM/'))
        for k in names:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        terms = []
        for s, z in zip(s_list, z_list):
            zr = M.modality_encoder_mu(s, self.decoder(s, z, self.P), self.P)[0]
            terms.append(O.mae(z, zr))
        total = self.conf['w_rec_Z'] * sum(terms)
        grads = torch.autograd.grad(total, [self.P[k] for k in names], allow_unused=True)
        self.last_grads = dict(zip(names, grads))
        self.adam['zreg'].step(self.P, self.last_grads)
        for k in names:
            self.P[k] = self.P[k].detach()
        return {'loss': float(total.detach())}

    @torch.no_grad()
    def mask_pool(self, *x):
        """4B fake masks: m(s1), m(s2), m(s1_def), m(s1_fused) (mmsdnet_executor.py:318-324); M modalities: m(s_i) of every
        modality + the (deformed, fused) masks of the pairs (0 -> j)."""
        P, nm = self.P, self.conf['num_masks']
        s = [self.enc(xi, i) for i, xi in enumerate(x)]
        ms = [M.segmentor(a, P, False, None) for a in s]
        for j in range(1, len(s)):
            ms += [M.segmentor(a, P, False, None) for a in M.anatomy_fuser(s[0], s[j], P)]
        return torch.cat(ms, 0)[..., :nm]

    def discriminator_step(self, real, fake):
        names = M.trainable_names(self.P, ('DM/',))
        for k in names:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        d_real = M.discriminator(real, self.P, 'DM/')
        d_fake = M.discriminator(fake, self.P, 'DM/')
        l_real, l_fake = O.mse(torch.ones_like(d_real), d_real), O.mse(torch.zeros_like(d_fake), d_fake)
        reg = M.discriminator_reg(self.P, 'DM/')
        total = l_real + l_fake + reg
        grads = torch.autograd.grad(total, [self.P[k] for k in names], allow_unused=True)
        self.last_grads = dict(zip(names, grads))
        self.adam['DM'].step(self.P, self.last_grads)
        for k in names:
            self.P[k] = self.P[k].detach()
        return {'loss': float(total.detach()), 'D_Mask_loss': float(l_fake.detach())}

    def train_batch(self, d):
        """MMSDNetExecutor.train_batch, l_mix = 1 (mmsdnet_executor.py:238-331): generator fit, Z_Regressor fit,
        D_Mask fit.  d: x1, x2, m1, m2 (4 ch), eps[6], z[6], dm_m, dm_x1, dm_x2, dm_idx."""
        res = OrderedDict()
        h = self.generator_step(d['x1'], d['x2'], d['m1'], d['m2'], d['eps'], True)
        res.update(supervised_Mask=h['Segmentor_loss'], adv_M=h['D_Mask_loss'], rec_X=h['Decoder_loss'], KL=h['Enc_Modality_loss'])
        res['rec_Z'] = self.zreg_step(self.zreg_inputs(d['x1'], d['x2']), d['z'])['loss']
        pool = self.mask_pool(d['dm_x1'], d['dm_x2'])
        res['dis_M'] = self.discriminator_step(d['dm_m'], pool[d['dm_idx']])['D_Mask_loss']
        return res
