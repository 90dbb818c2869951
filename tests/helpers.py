"""Shared helpers of the model-level tests: tiny configs, seeded synthetic batches, product -> oracle
parameter export."""
import numpy as np
import torch

from multimodal_segmentation_amd.utils.config import EasyDict


def make_conf(config_module, H, W=None, **overrides):
    conf = config_module.get()
    W = W or H
    shp = (H, W, 1)
    conf['input_shape'] = shp
    conf['anatomy_encoder']['input_shape'] = shp
    conf['anatomy_encoder']['output_shape'] = (H, W, conf['anatomy_encoder']['out_channels'])
    conf['d_mask_params']['input_shape'] = (H, W, conf['num_masks'])
    if 'd_image_params' in conf:
        conf['d_image_params']['input_shape'] = shp
    conf['n_pairs'] = 1
    conf['folder'] = '/tmp/mmseg_test_' + conf['folder']
    conf.update(overrides)
    return EasyDict(conf)


def smooth_field(rng, B, H, W, sigma=4.0):
    """tanh of low-pass filtered Gaussian noise, rescaled per slice to exactly [-1, 1] (mirrors chaos.py:242-246)."""
    from scipy.ndimage import gaussian_filter
    out = np.zeros((B, H, W, 1), np.float32)
    for b in range(B):
        f = np.tanh(gaussian_filter(rng.standard_normal((H, W)), sigma) * 6.0)
        f = (f - f.min()) / (f.max() - f.min() + 1e-12)
        out[b, ..., 0] = f * 2 - 1
    return out


def ellipse_masks(rng, B, H, W, num_masks=4):
    """num_masks disjoint random ellipses per slice -> [B,H,W,num_masks] in {0,1}."""
    yy, xx = np.mgrid[:H, :W]
    out = np.zeros((B, H, W, num_masks), np.float32)
    for b in range(B):
        taken = np.zeros((H, W), bool)
        for k in range(num_masks):
            cy, cx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W
            ry, rx = rng.uniform(0.06, 0.18) * H, rng.uniform(0.06, 0.18) * W
            m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0) & ~taken
            taken |= m
            out[b, ..., k] = m
    return out


def add_residual(data):
    residual = np.ones(data.shape[:-1] + (1,), np.float32)
    for i in range(data.shape[-1]):
        residual[data[..., i:i + 1] == 1] = 0
    return np.concatenate([data, residual], axis=-1).astype(np.float32)


def make_step_data(B, H, W, seed=1234, num_z=8):
    """Every tensor one DAFNet iteration consumes, incl. all random draws (see oracle/dafnet.py::train_batch)."""
    rng = np.random.RandomState(seed)
    d = {}
    for pre in ('', 'dm_', 'di_'):
        d[pre + 'x1'] = smooth_field(rng, B, H, W)
        d[pre + 'x2'] = smooth_field(rng, B, H, W)
    m = ellipse_masks(rng, B, H, W)
    d['m1'] = add_residual(m)
    d['m2'] = add_residual(ellipse_masks(rng, B, H, W))
    d['dm_m1'] = ellipse_masks(rng, B, H, W)
    d['dm_m2'] = ellipse_masks(rng, B, H, W)
    for k in ('z1', 'z2', 'eps1', 'eps2', 'di_eps1', 'di_eps2'):
        d[k] = rng.standard_normal((B, num_z)).astype(np.float32)
    d['dm_idx1'] = rng.choice(2 * B, B, replace=False)
    d['dm_idx2'] = rng.choice(2 * B, B, replace=False)
    d['di_idx1'] = rng.choice(3 * B, B, replace=False)
    d['di_idx2'] = rng.choice(3 * B, B, replace=False)
    return d


def to_torch(d, dtype):
    out = {}
    for k, v in d.items():
        out[k] = torch.as_tensor(v) if 'idx' in k else torch.as_tensor(v, dtype=dtype)
    return out


def export_dafnet(model, dtype=torch.float64):
    """product DAFNet -> oracle parameter dict (oracle/models.py naming)."""
    P = {}
    items = [('DM/', model.D_Mask), ('DI1/', model.D_Image1), ('DI2/', model.D_Image2),
             ('EA0/', model.Encoders_Anatomy[0]), ('EA1/', model.Encoders_Anatomy[1]),
             ('EAS/', model.Encoders_Anatomy[0].shared[0]), ('FUS/', model.Anatomy_Fuser), ('EM/', model.Enc_Modality),
             ('SEG/', model.Segmentor), ('DEC/', model.Decoder)]
    if getattr(model, 'Balancer', None) is not None:
        items.append(('BAL/', model.Balancer))
    for prefix, m in items:
        for k, v in m.named_weights(prefix).items():
            P[k] = torch.as_tensor(v, dtype=dtype)
    return P


def product_grads(model):
    """name -> gradient (numpy) of every trainable generator/discriminator weight, oracle naming."""
    out = {}
    items = [('DM/', model.D_Mask), ('DI1/', model.D_Image1), ('DI2/', model.D_Image2),
             ('EA0/', model.Encoders_Anatomy[0]), ('EA1/', model.Encoders_Anatomy[1]),
             ('EAS/', model.Encoders_Anatomy[0].shared[0]), ('FUS/', model.Anatomy_Fuser), ('EM/', model.Enc_Modality),
             ('SEG/', model.Segmentor), ('DEC/', model.Decoder)]
    if getattr(model, 'Balancer', None) is not None:
        items.append(('BAL/', model.Balancer))
    for prefix, m in items:
        for p in m.params.values():
            if p.trainable:
                out[prefix + p.name] = p.grad.detach().cpu().numpy().copy()
    return out


def _mmsdnet_items(model):
    return [('DM/', model.D_Mask)] + [('EA%d/' % i, e) for i, e in enumerate(model.Encoders_Anatomy)] + [
            ('FUS/', model.Anatomy_Fuser), ('EM/', model.Enc_Modality), ('SEG/', model.Segmentor), ('DEC/', model.Decoder)]


def export_mmsdnet(model, dtype=torch.float64):
    P = {}
    for prefix, m in _mmsdnet_items(model):
        for k, v in m.named_weights(prefix).items():
            P[k] = torch.as_tensor(v, dtype=dtype)
    return P


def product_grads_mmsdnet(model):
    out = {}
    for prefix, m in _mmsdnet_items(model):
        for p in m.params.values():
            if p.trainable:
                out[prefix + p.name] = p.grad.detach().cpu().numpy().copy()
    return out


# ---- free-running iterations: product trainers <-> oracle state -------------------------------------------------------------------
def dafnet_items(model):
    """(oracle prefix, product nn.Model) of every DAFNet component (the encoders' shared up-path once)"""
    items = [('DM/', model.D_Mask), ('DI1/', model.D_Image1), ('DI2/', model.D_Image2),
             ('EA0/', model.Encoders_Anatomy[0]), ('EA1/', model.Encoders_Anatomy[1]),
             ('EAS/', model.Encoders_Anatomy[0].shared[0]), ('FUS/', model.Anatomy_Fuser), ('EM/', model.Enc_Modality),
             ('SEG/', model.Segmentor), ('DEC/', model.Decoder)]
    if getattr(model, 'Balancer', None) is not None:
        items.append(('BAL/', model.Balancer))
    return items


DAFNET_TRAINERS = (('sup', 'supervised_trainer'), ('unsup', 'unsupervised_trainer'), ('DM', 'D_Mask_trainer'),
                   ('DI1', 'D_Image1_trainer'), ('DI2', 'D_Image2_trainer'))


def dafnet_targets(d, supervised=True):
    """targets of train_(un)supervised_expert_pairing (dafnet_executor.py:404-435) for the batch dict of make_step_data"""
    seg = [d['m1'], d['m2'], d['m1'], d['m2']] if supervised else [d['m1'], d['m1']]
    return seg + [1.0] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [1.0] * 4 + [0.0] * 2 + [d['z1'], d['z2']]


def product_train_batch(model, ex, d, supervised=True):
    """one FREE-RUNNING iteration of DAFNetExecutor.train_batch's schedule (dafnet_executor.py:369-387) on the product with every
    random draw taken from `d` (the arguments oracle.dafnet.DAFNetOracle.train_batch takes): generator fit, fake-mask pools + 2
    D_Mask fits, fake-image pools + D_Image1 / D_Image2 fits.  No teacher forcing.  -> History of the generator fit"""
    from multimodal_segmentation_amd import nn
    dev = lambda a: nn.to_device(a, model.D_Mask.device)
    sel = lambda pool, idx: pool.index_select(0, torch.as_tensor(np.asarray(idx), dtype=torch.long, device=pool.device))
    tr = model.supervised_trainer if supervised else model.unsupervised_trainer
    h = tr.fit([d['x1'], d['x2'], d['z1'], d['z2']], dafnet_targets(d, supervised), eps=[d['eps1'], d['eps2']])
    p1, p2 = ex.mask_pools(dev(d['dm_x1']), dev(d['dm_x2']))
    model.D_Mask_trainer.fit([d['dm_m1'], sel(p1, d['dm_idx1'])], [1.0, 0.0])
    model.D_Mask_trainer.fit([d['dm_m2'], sel(p2, d['dm_idx2'])], [1.0, 0.0])
    y1, y2 = ex.image_pools(dev(d['di_x1']), dev(d['di_x2']), d['di_eps1'], d['di_eps2'])
    model.D_Image1_trainer.fit([d['di_x1'], sel(y1, d['di_idx1'])], [1.0, 0.0])
    model.D_Image2_trainer.fit([d['di_x2'], sel(y2, d['di_idx2'])], [1.0, 0.0])
    return h


def product_state(model):
    """-> (weights + BatchNorm moving statistics by oracle name, {adam key: (t, {name: m}, {name: v})}) as float64 numpy"""
    W, A = {}, {}
    items = dafnet_items(model)
    for prefix, m in items:
        for p in m.params.values():
            W[prefix + p.name] = p.data.detach().double().cpu().numpy().copy()
    for key, attr in DAFNET_TRAINERS:
        opt = getattr(model, attr).optimizer
        ms, vs = {}, {}
        for prefix, m in items:
            st = opt.state.get(m.uid)
            if st is None:
                continue
            for p in m.params.values():
                if p.trainable:
                    ms[prefix + p.name] = st[0][p.offset:p.offset + p.numel].detach().double().cpu().numpy().reshape(p.shape)
                    vs[prefix + p.name] = st[1][p.offset:p.offset + p.numel].detach().double().cpu().numpy().reshape(p.shape)
        A[key] = (opt.iterations, ms, vs)
    return W, A


def load_oracle_state(model, orc):
    """transplant the oracle's complete training state into the product: weights, BatchNorm moving statistics, and the Adam
    iteration count / first / second moments of every trainer (so that the NEXT iteration is the same map on both sides)"""
    from multimodal_segmentation_amd import ops
    items = dafnet_items(model)
    for prefix, m in items:
        for p in m.params.values():
            p.data.copy_(orc.P[prefix + p.name].detach().to(torch.float32).reshape(p.shape).to(p.data.device))
    ops.bump_weight_version()
    for key, attr in DAFNET_TRAINERS:
        opt, oad = getattr(model, attr).optimizer, orc.adam[key]
        opt.iterations = oad.t
        opt.state = {}
        for prefix, m in items:
            names = [prefix + p.name for p in m.params.values() if p.trainable]
            if not any(n in oad.m for n in names):
                continue
            st = (torch.zeros_like(m.arena), torch.zeros_like(m.arena))
            for p in m.params.values():
                n = prefix + p.name
                if p.trainable and n in oad.m:
                    st[0][p.offset:p.offset + p.numel].copy_(oad.m[n].detach().to(torch.float32).reshape(-1).to(m.arena.device))
                    st[1][p.offset:p.offset + p.numel].copy_(oad.v[n].detach().to(torch.float32).reshape(-1).to(m.arena.device))
            opt.state[m.uid] = st


def sharpen_anatomy_heads(model, factor=40.0, theta_std=0.002, seed=3):
    """make the freshly initialised model non-trivial for a free-running comparison: scale the anatomy encoders' 1x1 softmax head
    so that the ROUNDED anatomies are not all zero (at initialisation every softmax channel sits near 1/8), and move the
    zero-initialised theta layer of the fuser off the identity warp"""
    rng = np.random.RandomState(seed)
    shared = model.Encoders_Anatomy[0].shared[0]
    names = [n for n in shared.params if n.endswith('/kernel')]
    head = shared.params[names[-1]]
    head.data.mul_(factor)
    th = model.Anatomy_Fuser.params['theta/kernel']
    th.data.copy_(torch.from_numpy((rng.standard_normal(th.shape) * theta_std).astype(np.float32)).to(th.data.device))
    from multimodal_segmentation_amd import ops
    ops.bump_weight_version()
    return head.name


# ---- teacher forcing across the Rounding discontinuity ------------------------------------------------------------
class _SteReplace(torch.autograd.Function):
    """replace a rounded anatomy by a given tensor while passing the gradient straight through (no arithmetic)"""

    @staticmethod
    def forward(ctx, s, teacher):
        return teacher.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


class teacher_forcing(object):
    """`with teacher_forcing(model, teacher): trainer.fit(...)` -- every anatomy encoder of `model` returns the given anatomies instead
    of its own rounded output (straight-through gradient) while the block runs.  `teacher`: per modality one tensor, or a list of
    tensors for the successive calls of that modality's encoder (automated pairing: one per candidate slice); None = no forcing.
    Installed through model_components.anatomy_encoder.set_rounding_hook: the product graphs carry no test arguments."""

    def __init__(self, model, teacher):
        self.model, self.teacher = model, teacher

    def __enter__(self):
        from multimodal_segmentation_amd.model_components import anatomy_encoder as AE
        self.AE = AE
        if self.teacher is None:
            self.prev = AE.set_rounding_hook(AE._rounding_hook[0])
            return self
        queues = {id(enc): (list(t) if isinstance(t, (list, tuple)) else [t]) for enc, t in zip(self.model.Encoders_Anatomy, self.teacher)}

        def hook(enc, s):
            q = queues.get(id(enc))
            return _SteReplace.apply(s, q.pop(0)) if q else s
        self.prev = AE.set_rounding_hook(hook)
        return self

    def __exit__(self, *exc):
        self.AE.set_rounding_hook(self.prev)
        return False
