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
