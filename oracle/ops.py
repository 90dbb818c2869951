"""Oracle ops: torch-CPU restatement of every op on the DAFNet/MMSDNet hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All tensors are NHWC like the
reference.  Differentiation is left to torch-CPU autograd, except where the
reference defines its own gradient (Rounding straight-through, stop_gradient in
the spectral regulariser).  Citations are relative to /root/reference.
A trailing (dd) marks a Keras/TF/keras-contrib default restated from memory.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.99   # keras.layers.BatchNormalization default (dd)
BN_EPS = 1e-3        # keras.layers.BatchNormalization default (dd)
LRELU_DEFAULT = 0.3  # keras.layers.LeakyReLU() default alpha (dd)


# ----------------------------------------------------------------------------
# convolution / dense (Keras Conv2D, Dense; weights in Keras layout)
# ----------------------------------------------------------------------------
_OPERAND_ROUNDING = [None]


def set_conv_operand_rounding(dtype):
    """None (default) | torch.bfloat16 | torch.float16: emulate the product's reduced-precision compute modes (BASELINE
    configs #3 / #5; NOT a reference behaviour -- the reference is fp32 throughout).  In those modes the product's MFMA
    convolutions (input channels a multiple of 4) multiply operands ROUNDED to the 16-bit
    type and accumulate in fp32; every other operation stays fp32.  With the rounding emulated here the products are exact in
    both implementations and only the accumulation order differs.  Returns the previous setting."""
    old = _OPERAND_ROUNDING[0]
    _OPERAND_ROUNDING[0] = dtype
    return old


def _round_operand(t):
    return t.to(_OPERAND_ROUNDING[0]).to(t.dtype)


def conv2d(x, w, b=None, stride=1, padding='same'):
    """Keras Conv2D on NHWC input with HWIO kernel.

    'same' here is only used with stride 1 and odd kernels in the reference
    (models/unet.py:95-99, model_components/*), so TF's SAME padding is the
    symmetric (k-1)/2.  'valid' = no padding (modality_encoder.py:38-45,
    stn_spline.py:106-114, discriminator.py:24,39).
    """
    kh, kw = w.shape[0], w.shape[1]
    if padding == 'same':
        assert stride == 1 and kh % 2 == 1 and kw % 2 == 1
        pad = (kh // 2, kw // 2)
    else:
        pad = (0, 0)
    # MFMA kernels of the product in those modes: every forward convolution whose input channel count is a multiple of 4,
    # except the 8 -> 8 3x3 'same' layers of the FiLM decoder (a direct fp32 FMA kernel)
    direct8 = (kh == 3 and kw == 3 and w.shape[2] == 8 and w.shape[3] == 8 and stride == 1 and padding == 'same')
    if _OPERAND_ROUNDING[0] is not None and w.shape[2] % 4 == 0 and not direct8:
        x, w = _round_operand(x), _round_operand(w)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1)


def dense(x, w, b=None):
    y = x @ w
    return y if b is None else y + b


def flatten(x):
    """keras.layers.Flatten on NHWC: row-major over (H, W, C)."""
    return x.reshape(x.shape[0], -1)


def leaky_relu(x, alpha=LRELU_DEFAULT):
    return torch.where(x >= 0, x, x * alpha)


# ----------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------
def batchnorm(x, P, prefix, training, updates=None):
    """keras BatchNormalization(axis=-1) (dd): momentum .99, eps 1e-3.

    training: batch statistics (biased variance) normalise; the moving mean /
    variance are updated as  mov -= (mov - stat) * (1 - momentum)  with the
    *unbiased* batch variance (tf.nn.fused_batch_norm semantics) (dd).
    inference (`predict`): moving statistics.
    `updates` collects (name, new_value) pairs; the caller applies them in call
    order after the optimiser step (Keras runs them inside the same session.run).
    Call sites: utils/model_utils.py:6-12, model_components/segmentor.py:17,20.
    """
    g, b = P[prefix + '/gamma'], P[prefix + '/beta']
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = x.var(dim=(0, 1, 2), unbiased=False)
        y = (x - mean) * torch.rsqrt(var + BN_EPS) * g + b
        if updates is not None:
            n = x.shape[0] * x.shape[1] * x.shape[2]
            updates.append((prefix + '/moving_mean', mean.detach()))
            updates.append((prefix + '/moving_variance', (var * (n / max(n - 1, 1))).detach()))
        return y
    mm, mv = P[prefix + '/moving_mean'], P[prefix + '/moving_variance']
    return (x - mm) * torch.rsqrt(mv + BN_EPS) * g + b


def apply_bn_updates(P, updates):
    """Sequential moving-average updates in call order (shared BN layers are
    called once per modality: model_components/anatomy_encoder.py:48-51)."""
    for name, stat in updates:
        P[name] = P[name] - (P[name] - stat) * (1.0 - BN_MOMENTUM)


def instance_norm(x, eps=1e-3):
    """keras_contrib InstanceNormalization(axis=None, scale=False, center=False) (dd):
    statistics over (H, W, C) jointly per sample, (x - mean) / (std + eps).
    Call site: layers/spade.py:27."""
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    std = x.var(dim=(1, 2, 3), unbiased=False, keepdim=True).sqrt() + eps
    return (x - mean) / std


# ----------------------------------------------------------------------------
# resampling
# ----------------------------------------------------------------------------
def maxpool2(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)


def upsample2(x):
    """keras UpSampling2D(size=2): nearest (repeat)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def resize_nearest(x, H, W):
    """tf.image.resize_nearest_neighbor, align_corners=False (dd):
    src = min(floor(dst * in/out), in-1).  layers/spade.py:36-38."""
    h, w = x.shape[1], x.shape[2]
    ri = torch.clamp(torch.floor(torch.arange(H, dtype=torch.float64) * (h / H)).long(), max=h - 1)
    ci = torch.clamp(torch.floor(torch.arange(W, dtype=torch.float64) * (w / W)).long(), max=w - 1)
    return x[:, ri][:, :, ci]


# ----------------------------------------------------------------------------
# Rounding with straight-through gradient (layers/rounding.py:23-42)
# ----------------------------------------------------------------------------
class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return torch.round(x)  # half-to-even, like np.round (rounding.py:35)

    @staticmethod
    def backward(ctx, g):
        return g * 1  # rounding.py:40-42


def round_ste(x):
    return _RoundSTE.apply(x)


# ----------------------------------------------------------------------------
# FiLM / SPADE
# ----------------------------------------------------------------------------
def film(x, gamma, beta):
    """layers/film.py:26-36: x * gamma + beta, gamma/beta [B, C] tiled over H, W."""
    return x * gamma[:, None, None, :] + beta[:, None, None, :]


def spade_cond(x, gamma, beta):
    """layers/spade.py:51-54."""
    return x * (1 + gamma) + beta


# ----------------------------------------------------------------------------
# thin-plate spline + bilinear resampler
# ----------------------------------------------------------------------------
def nd_grid(dims, dtype):
    """layers/stn_spline.py:70-91 (2-D, normalise=True): rows of (row/(H-1), col/(W-1))."""
    g = np.mgrid[:dims[0], :dims[1]].reshape((2, -1)).T[None]
    g = g / (1. * (np.array([[dims]]) - 1))
    return torch.as_tensor(g, dtype=dtype)


def _phi(r):
    """layers/interpolate_spline.py:198-199 (order 2)."""
    return 0.5 * r * torch.log(torch.clamp(r, min=1e-10))


def _cross_sq_dist(x, y):
    """layers/interpolate_spline.py:30-51."""
    xn = (x * x).sum(2)
    yn = (y * y).sum(2)
    return xn[:, :, None] - 2 * (x @ y.transpose(1, 2)) + yn[:, None, :]


def _pairwise_sq_dist(x):
    """layers/interpolate_spline.py:54-73."""
    xxt = x @ x.transpose(1, 2)
    xn = torch.diagonal(xxt, dim1=1, dim2=2)[:, :, None]
    return xn - 2 * xxt + xn.transpose(1, 2)


def interpolate_spline(train_points, train_values, query_points):
    """layers/interpolate_spline.py:76-179,212-278 with order=2, regularization 0."""
    c, f = train_points, train_values
    b, n, d = c.shape
    k = f.shape[-1]
    A = _phi(_pairwise_sq_dist(c))
    Bm = torch.cat([c, torch.ones_like(c[..., :1])], 2)
    left = torch.cat([A, Bm.transpose(1, 2)], 1)
    right = torch.cat([Bm, torch.zeros(b, d + 1, d + 1, dtype=c.dtype, device=c.device)], 1)
    lhs = torch.cat([left, right], 2)
    rhs = torch.cat([f, torch.zeros(b, d + 1, k, dtype=c.dtype, device=c.device)], 1)
    wv = torch.linalg.solve(lhs, rhs)
    w, v = wv[:, :n], wv[:, n:]
    rbf = _phi(_cross_sq_dist(query_points, c)) @ w
    qp = torch.cat([query_points, torch.ones_like(query_points[..., :1])], 2)
    return rbf + qp @ v


def resampler(data, warp):
    """tf.contrib.resampler.resampler (dd): data [B,H,W,C], warp [B,N,2] as (x, y)
    in pixel units; bilinear, taps outside the image contribute 0; differentiable
    w.r.t. data and warp (autograd of the formula below equals the op's
    registered gradient).  Call site layers/stn_spline.py:8,65."""
    B, H, W, C = data.shape
    x, y = warp[..., 0], warp[..., 1]
    fx, fy = torch.floor(x), torch.floor(y)
    ax, ay = x - fx, y - fy           # weight of the ceil tap
    fxi, fyi = fx.long(), fy.long()
    flat = data.reshape(B, H * W, C)

    def tap(xi, yi):
        ok = ((xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1))
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1))
        v = torch.gather(flat, 1, idx[..., None].expand(-1, -1, C))
        return v * ok[..., None].to(data.dtype)

    out = ((1 - ax) * (1 - ay))[..., None] * tap(fxi, fyi) + (ax * (1 - ay))[..., None] * tap(fxi + 1, fyi) \
        + ((1 - ax) * ay)[..., None] * tap(fxi, fyi + 1) + (ax * ay)[..., None] * tap(fxi + 1, fyi + 1)
    return out


def tps_warp(vol, theta, cp_dims=(5, 5)):
    """ThinPlateSpline2D.call, inverse=False, order=2 (layers/stn_spline.py:36-67)."""
    B, H, W, C = vol.shape
    cp = nd_grid(cp_dims, vol.dtype).to(vol.device)          # [1, 25, 2]
    q = nd_grid((H, W), vol.dtype).to(vol.device)            # [1, HW, 2]
    locs = []
    for i in range(B):                        # tf.map_fn over the batch (stn_spline.py:58)
        locs.append(interpolate_spline(cp, cp + theta[i][None], q)[0])
    loc = torch.stack(locs, 0)
    loc = torch.flip(loc, dims=[-1])          # (row, col) -> (x, y)   (stn_spline.py:60)
    loc = loc * torch.tensor([W - 1, H - 1], dtype=vol.dtype, device=vol.device)  # stn_spline.py:62-63
    return resampler(vol, loc).reshape(B, H, W, C)


def tps_basis(H, W, cp_dims=(5, 5)):
    """Constant M [HW, 25] (fp64) with  loc_normalised = grid + M @ theta.

    Not in the reference: a closed form of interpolate_spline for the fixed 5x5
    control grid used by anatomy_fuser.py:24,30 (inverse=False), used to cross-
    check tps_warp and by the product kernel."""
    cp = nd_grid(cp_dims, torch.float64)
    q = nd_grid((H, W), torch.float64)
    n = cp.shape[1]
    eye = torch.eye(n, dtype=torch.float64)[None]
    return interpolate_spline(cp, eye, q)[0]


# ----------------------------------------------------------------------------
# spectral-norm regulariser (layers/spectralnorm.py:199-239)
# ----------------------------------------------------------------------------
def spectral_reg(w, u0, alpha=10.0):
    """3 power iterations from the *initial* u0 every call (self.u is re-bound,
    never assigned back: spectralnorm.py:228-234); penalty
    alpha * mean|stop_gradient(W / sigma) - W| (236-239)."""
    x = w.reshape(-1, w.shape[-1])
    u = u0
    for _ in range(3):
        wtu = x.t() @ u
        v = wtu / torch.sqrt((wtu * wtu).sum())
        wv = x @ v
        u = wv / torch.sqrt((wv * wv).sum())
    sigma = (u.t() @ x) @ v
    target = (x / sigma).detach()
    return alpha * (target - x).abs().mean()


# ----------------------------------------------------------------------------
# losses (costs.py)
# ----------------------------------------------------------------------------
LAMBDA_BCE = 0.01  # costs.py:10


def dice_loss(y_true, y_pred, restrict_chn):
    """make_dice_loss_fnc / dice_coef_loss / dice_coef_perbatch (costs.py:43-67)."""
    t = y_true[..., :restrict_chn]
    p = y_pred[..., :restrict_chn]
    inter = (t * p).sum(dim=(1, 2, 3))
    union = t.sum(dim=(1, 2, 3)) + p.sum(dim=(1, 2, 3))
    return (1 - (2 * inter + 1e-12) / (union + 1e-12)).mean()


def weighted_cross_entropy_loss(y_pred, y_true):
    """costs.py:70-85, parameter names as DECLARED there."""
    nc = y_true.shape[-1]
    n = y_true.sum(dim=(0, 1, 2))
    n_tot = n.sum()
    weights = n_tot / (n + 1e-12)
    yp = y_pred.reshape(-1, nc)
    yt = y_true.reshape(-1, nc)
    wce = -(yt * torch.log(yp + 1e-12) * weights).sum(1)
    return wce.mean()


def combined_dice_bce(y_true, y_pred, num_classes):
    """make_combined_dice_bce (costs.py:129-136).  NOTE the call
    bce(y_true, y_pred) into a function declared (y_pred, y_true): the class
    weights come from the predictions and the log is taken of the labels."""
    return dice_loss(y_true, y_pred, num_classes) + LAMBDA_BCE * weighted_cross_entropy_loss(y_true, y_pred)


def kl(mean, log_var):
    """costs.py:186-189 -> [B, 1]."""
    return (-0.5 * (1 + log_var - mean * mean - torch.exp(log_var)).sum(-1)).reshape(-1, 1)


def sampling(z_mean, z_log_var, eps):
    """utils/sdnet_utils.py:9-21 with the in-graph N(0, I) draw made explicit."""
    return z_mean + torch.exp(0.5 * z_log_var) * eps


def mae(y_true, y_pred):
    return (y_pred - y_true).abs().mean()   # keras 'mae' then batch mean (dd)


def mse(y_true, y_pred):
    return ((y_pred - y_true) ** 2).mean()  # keras 'mse' then batch mean (dd)


def dice_metric(y_true, y_pred, binarise=False, smooth=1e-12):
    """numpy metric costs.py:31-41."""
    y_pred = y_pred[..., 0:y_true.shape[-1]]
    if binarise:
        y_pred = np.round(y_pred)
    y_int = y_true * y_pred
    return np.mean((2 * np.sum(y_int, axis=(1, 2, 3)) + smooth)
                   / (np.sum(y_true, axis=(1, 2, 3)) + np.sum(y_pred, axis=(1, 2, 3)) + smooth))


# ----------------------------------------------------------------------------
# Keras 2.1.6 Adam (dd): eps = K.epsilon() = 1e-7 outside the sqrt, bias
# correction folded into lr_t.  Call sites models/dafnet.py:93,114,155,161,349.
# ----------------------------------------------------------------------------
class KerasAdam(object):
    def __init__(self, lr, beta_1=0.9, beta_2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, P, grads):
        """P: dict name -> tensor (updated in place by re-binding), grads: dict name -> tensor."""
        self.t += 1
        lr_t = self.lr * math.sqrt(1. - self.b2 ** self.t) / (1. - self.b1 ** self.t)
        for k, g in grads.items():
            if g is None:
                continue
            m = self.m.get(k, torch.zeros_like(g))
            v = self.v.get(k, torch.zeros_like(g))
            m = self.b1 * m + (1. - self.b1) * g
            v = self.b2 * v + (1. - self.b2) * g * g
            P[k] = (P[k] - lr_t * m / (torch.sqrt(v) + self.eps)).detach()
            self.m[k], self.v[k] = m, v


def add_residual(data):
    """model_executors/base_executor.py:83-87 (numpy)."""
    residual = np.ones(data.shape[:-1] + (1,))
    for i in range(data.shape[-1]):
        residual[data[..., i:i + 1] == 1] = 0
    return np.concatenate([data, residual], axis=-1)


# ----------------------------------------------------------------------------
# in-graph per-sample losses of the automated-pairing trainers
# ----------------------------------------------------------------------------
def pair_dice(a, b):
    """model_components/balancer.py:33-38 -> [B, 1]"""
    inter = (a * b).sum(dim=(1, 2, 3))
    union = a.sum(dim=(1, 2, 3)) + b.sum(dim=(1, 2, 3))
    return ((2 * inter + 1e-12) / (union + 1e-12)).unsqueeze(1)


def mae_single_input(y1, y2):
    """costs.py:24-26: K.mean(|y1 - y2|, axis=(1, 2)) -> [B, C] ([B, 1] for images)"""
    return (y1 - y2).abs().mean(dim=(1, 2))


def dice_coef_perbatch(y_true, y_pred):
    """costs.py:43-49 -> [B]"""
    inter = (y_true * y_pred).sum(dim=(1, 2, 3))
    union = y_true.sum(dim=(1, 2, 3)) + y_pred.sum(dim=(1, 2, 3))
    return 1 - (2 * inter + 1e-12) / (union + 1e-12)


def weighted_cross_entropy_perbatch(y_pred, y_true):
    """costs.py:88-108, with the parameter names of its declaration.  NOTE the only caller (costs.py:142) passes
    (y_true, y_pred), i.e. swapped: class weights then come from the prediction, softmax + log is applied to the labels."""
    B, H, W, C = y_true.shape
    n = y_true.sum(dim=(0, 1, 2))
    weights = n.sum() / (n + 1e-12)
    sm = torch.softmax(y_pred.reshape(B, H * W, C), dim=-1)
    ce = -(y_true.reshape(B, H * W, C) * torch.log(sm + 1e-12) * weights).sum(dim=2)
    return ce.mean(dim=1)


def combined_dice_bce_perbatch(y_true, y_pred, num_classes):
    """costs.py:138-143 -> [B]"""
    return dice_coef_perbatch(y_true[..., :num_classes], y_pred[..., :num_classes]) + \
        0.01 * weighted_cross_entropy_perbatch(y_true, y_pred)
