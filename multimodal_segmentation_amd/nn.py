"""Minimal Keras-shaped model layer for the DAFNet/MMSDNet engine.

What the reference's callers touch of `keras.Model` is mirrored here: calling a model on tensors,
`.predict`, `.get_weights/.set_weights`, `.save_weights/.load_weights`, `.summary`, `.name`, `.trainable`,
`.layers`, `.get_layer`, `.input_shape/.output_shape`.  Everything is define-by-run on device tensors.

Memory layout (HBM): every Model keeps its trainable weights in ONE flat fp32 arena (`model.arena`), its
gradients in a second arena of the same layout (`model.grad_arena`) and its non-trainable state (BatchNorm moving
statistics, the spectral regulariser's u0) in a third.  Parameter tensors are 16-byte aligned views.  The Adam
kernel and the data-parallel all-reduce therefore work on whole arenas (one launch / one collective per model).
Weights are NOT autograd leaves: the autograd tape only carries activation gradients, and the kernels accumulate
weight gradients straight into the gradient arena.
"""
import itertools
import math
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .parallel import dp

_DEVICE = None


def default_device():
    global _DEVICE
    if _DEVICE is None:
        if torch.cuda.is_available():
            _DEVICE = torch.device('cuda', torch.cuda.current_device())
        else:
            _DEVICE = torch.device('cpu')   # only reachable under the test stand-in; kernels refuse CPU tensors
    return _DEVICE


def set_default_device(dev):
    global _DEVICE
    _DEVICE = torch.device(dev)


_anchors = {}


def anchor(device):
    """A 1-element tensor that requires grad: fed to every node with trainable weights so that autograd visits
    the node even when none of its activations requires grad (first layer of a network)."""
    a = _anchors.get(device)
    if a is None:
        a = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        _anchors[device] = a
    return a


# ---- initialisers (Keras 2.1.6 defaults: he_normal/glorot_normal are truncated normals) -----------------------
def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2
    return out * std


def initialise(rng, shape, kind):
    if kind == 'zeros':
        return np.zeros(shape, np.float32)
    if kind == 'ones':
        return np.ones(shape, np.float32)
    if kind == 'uniform_pm1':      # Spectral.u (layers/spectralnorm.py:213 of the reference)
        return (rng.random_sample(shape) * 2 - 1.).astype(np.float32)
    fi, fo = _fans(shape)
    if kind == 'he_normal':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / fi)).astype(np.float32)
    if kind == 'glorot_normal':
        return _trunc_normal(rng, shape, math.sqrt(2.0 / (fi + fo))).astype(np.float32)
    if kind == 'glorot_uniform':
        lim = math.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    raise ValueError(kind)


_serial = itertools.count(1)      # uids of models and parameters: cache keys that are never reused (unlike id())


class Param(object):
    __slots__ = ('name', 'shape', 'init', 'trainable', 'data', 'grad', 'owner', 'offset', 'uid', 'seg')

    def __init__(self, name, shape, init, trainable=True):
        self.name, self.shape, self.init, self.trainable = name, tuple(shape), init, trainable
        self.uid = next(_serial)
        self.data = self.grad = self.owner = None
        self.offset = 0
        self.seg = 0           # segment of the owner's gradient arena this parameter lies in (Model.grad_segments)

    @property
    def numel(self):
        return int(np.prod(self.shape))

    def g(self):
        """Gradient view to accumulate into, or None when the owning model is frozen / not recording."""
        if not (self.trainable and self.owner.trainable and torch.is_grad_enabled()):
            return None
        tr = dp.current_tracker()
        if tr is not None:                 # data parallel: count the launches that will accumulate into this arena segment
            tr.register(self.owner, self.seg)
        return self.grad


class Model(object):
    """A named component with parameters.  Sub-classes declare parameters in __init__ via add_param (Keras weight
    order = declaration order), call finalize(), and implement forward(*tensors, training=...)."""

    def __init__(self, name):
        self.name = name
        self.uid = next(_serial)
        weakref.finalize(self, ops.evict_owner, self.uid)      # the cached images of a dead model leave HBM with it
        self.trainable = True
        self.params = OrderedDict()
        self.arena = self.grad_arena = self.state_arena = None
        self.grad_segments = [(0, 0)]      # contiguous [start, end) ranges of grad_arena that are all-reduced separately (parallel/dp.py)
        self.input_shape = self.output_shape = None
        self.shared = []           # other Models whose weights are part of this one (shared layers)
        self.precision = None      # (compute_dtype, 16-bit activation storage) of the model wrapper that built it; None: inherit

    # ---- construction --------------------------------------------------------------------------------------
    def add_param(self, name, shape, init, trainable=True):
        p = Param(name, shape, init, trainable)
        p.owner = self
        self.params[name] = p
        return p

    def finalize(self, rng, device=None):
        device = device or default_device()
        off_t = off_s = 0
        for p in self.params.values():
            n = (p.numel + 3) // 4 * 4     # 16-byte aligned views
            if p.trainable:
                p.offset, off_t = off_t, off_t + n
            else:
                p.offset, off_s = off_s, off_s + n
        # data-parallel segments: cut at layer boundaries (a '/kernel' that opens a new layer) once a segment holds >= SEGMENT_FLOATS
        segs, start, cur_layer = [], 0, None
        for p in self.params.values():
            if not p.trainable:
                continue
            layer = p.name.split('/')[0]
            if layer != cur_layer and p.name.endswith('/kernel') and p.offset - start >= dp.SEGMENT_FLOATS:
                segs.append((start, p.offset))
                start = p.offset
            cur_layer = layer
            p.seg = len(segs)
        segs.append((start, max(off_t, 4)))
        self.grad_segments = segs
        host_t = np.zeros(max(off_t, 4), np.float32)
        host_s = np.zeros(max(off_s, 4), np.float32)
        for p in self.params.values():
            v = initialise(rng, p.shape, p.init).reshape(-1)
            (host_t if p.trainable else host_s)[p.offset:p.offset + p.numel] = v
        self.arena = torch.from_numpy(host_t).to(device)
        self.state_arena = torch.from_numpy(host_s).to(device)
        self.grad_arena = torch.empty_like(self.arena)
        ops.fill_(self.grad_arena, 0.0)
        for p in self.params.values():
            if p.trainable:
                p.data = self.arena[p.offset:p.offset + p.numel].view(p.shape)
                p.grad = self.grad_arena[p.offset:p.offset + p.numel].view(p.shape)
                p.grad._owner = self           # lets the autograd Functions report 'gradient queued' to the DP tracker
                p.grad._seg = p.seg
            else:
                p.data = self.state_arena[p.offset:p.offset + p.numel].view(p.shape)
        self.device = device
        return self

    # ---- keras.Model surface -------------------------------------------------------------------------------
    @property
    def layers(self):
        seen, out = set(), []
        for n in self.params:
            l = n.split('/')[0]
            if l not in seen:
                seen.add(l)
                out.append(l)
        return out

    def get_layer(self, name):
        ps = OrderedDict((k, v) for k, v in self.params.items() if k.split('/')[0] == name)
        if not ps:
            raise ValueError('No such layer: ' + name)
        return ps

    def all_params(self):
        ps = list(self.params.values())
        for m in self.shared:
            ps += m.all_params()
        return ps

    def owned_models(self):
        out = [self]
        for m in self.shared:
            out += m.owned_models()
        return out

    def get_weights(self):
        return [p.data.detach().cpu().numpy().copy() for p in self.all_params()]

    def set_weights(self, weights):
        ps = self.all_params()
        if len(weights) != len(ps):
            raise ValueError('%s expects %d weight arrays, got %d' % (self.name, len(ps), len(weights)))
        for p, w in zip(ps, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != p.shape:
                raise ValueError('weight %s: shape %s != %s' % (p.name, w.shape, p.shape))
            p.data.copy_(torch.from_numpy(w).to(p.data.device))
        ops.bump_weight_version()

    def named_weights(self, prefix=''):
        """name -> numpy array (the oracle's parameter naming)."""
        out = OrderedDict()
        for p in self.params.values():
            out[prefix + p.name] = p.data.detach().cpu().numpy().copy()
        return out

    def count_params(self):
        return sum(p.numel for p in self.all_params())

    def save_weights(self, path):
        arrs = {('%04d' % i): w for i, w in enumerate(self.get_weights())}
        with open(path, 'wb') as f:      # same file name as the reference (no extension), npz container
            np.savez(f, **arrs)

    def load_weights(self, path):
        with np.load(path) as z:
            self.set_weights([z[k] for k in sorted(z.files)])

    def summary(self, print_fn=print):
        print_fn('Model: %s' % self.name)
        for p in self.all_params():
            print_fn('  %-40s %-24s %d' % (p.name, str(p.shape), p.numel))
        print_fn('Total params: %d' % self.count_params())

    def get_output_shape_at(self, idx):
        return self.output_shape

    # ---- execution -----------------------------------------------------------------------------------------
    def forward(self, *inputs, **kw):
        raise NotImplementedError

    def __call__(self, *inputs, **kw):
        return self.forward(*inputs, **kw)

    def predict(self, inputs, **kw):
        """keras `predict`: inference mode (moving BN statistics), no tape.  numpy in -> numpy out; device tensors
        in -> device tensors out (no host round trip)."""
        single = not isinstance(inputs, (list, tuple))
        ins = [inputs] if single else list(inputs)
        as_numpy = not isinstance(ins[0], torch.Tensor)
        ts = [to_device(x, self.device) for x in ins]
        with torch.no_grad(), ops.precision_scope(self.precision):
            out = self.forward(*ts, training=False, **kw)
        if isinstance(out, (list, tuple)):
            return [to_numpy(o) for o in out] if as_numpy else list(out)
        return to_numpy(out) if as_numpy else out

    def zero_grad_own(self):
        """Zero this model's own gradient arena (shared sub-models are separate entries of a trainer's list)."""
        ops.fill_(self.grad_arena, 0.0)

    def zero_grad(self):
        for m in self.owned_models():
            ops.fill_(m.grad_arena, 0.0)


def host_to_device(a, device, dtype=np.float32):
    """numpy -> device tensor WITHOUT blocking the host: the array is staged in pinned memory (torch's caching host
    allocator recycles the block only after the copy has run) and copied asynchronously on the current stream.  A plain
    `.to(device)` from pageable memory makes the host wait until the GPU has drained everything queued before the copy --
    with ~25 small uploads per iteration (z samples, eps, pool indices, batch rows, rotation matrices) that empties the
    launch queue again and again."""
    a = np.ascontiguousarray(a, dtype=dtype)
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.from_numpy(a).to(device)
    src = torch.from_numpy(a)
    staged = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    staged.copy_(src)
    return staged.to(device, non_blocking=True)


def to_device(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32)
    return host_to_device(x, device, np.float32)


def to_numpy(t):
    return t.detach().cpu().numpy()


# ---- layer helpers: declare parameters with the oracle's names, apply with the fused kernels -------------------
def conv_params(m, name, k, cin, cout, init='glorot_uniform', bias=True):
    m.add_param(name + '/kernel', (k, k, cin, cout), init)
    if bias:
        m.add_param(name + '/bias', (cout,), 'zeros')


def dense_params(m, name, cin, cout, init='glorot_uniform'):
    m.add_param(name + '/kernel', (cin, cout), init)
    m.add_param(name + '/bias', (cout,), 'zeros')


def bn_params(m, name, c):
    m.add_param(name + '/gamma', (c,), 'ones')
    m.add_param(name + '/beta', (c,), 'zeros')
    m.add_param(name + '/moving_mean', (c,), 'zeros', trainable=False)
    m.add_param(name + '/moving_variance', (c,), 'ones', trainable=False)


def conv(m, name, x, stride=1, padding='same', act=None, alpha=0.0, x2=None, upsample=False, bias_grad=True, out_dtype=torch.float32):
    """bias_grad=False: the convolution feeds a training-mode BatchNorm, whose backward output sums to zero over
    (N,H,W) per channel, so the bias gradient is EXACTLY zero in exact arithmetic (the reference accumulates only
    rounding noise there); the column-sum pass over the gradient tensor is skipped and the bias keeps its value."""
    w = m.params[name + '/kernel']
    b = m.params.get(name + '/bias')
    return ops.conv2d(x, w.data, b.data if b is not None else None, stride, padding, act, alpha, x2, upsample,
                      wgrad=w.g(), bgrad=b.g() if (b is not None and bias_grad) else None, anchor=anchor(x.device),
                      wkey=(w.uid, w.owner.uid), out_dtype=out_dtype)


def conv_pair(m, name_a, name_b, x, out_dtype=torch.float32):
    """two 3x3 'same' convolutions of the same input as one launch with their output channels side by side (ops.conv2d_pair): the
    gamma and beta convolutions of a SPADE unit (reference layers/spade.py:30-31)"""
    wa, ba = m.params[name_a + '/kernel'], m.params[name_a + '/bias']
    wb, bb = m.params[name_b + '/kernel'], m.params[name_b + '/bias']
    grads = (wa.g(), ba.g(), wb.g(), bb.g())
    return ops.conv2d_pair(x, wa.data, ba.data, wb.data, bb.data, grads, anchor=anchor(x.device), wkey=(wa.uid, wa.owner.uid),
                           out_dtype=out_dtype)


def dense(m, name, x, act=None, alpha=0.0):
    w, b = m.params[name + '/kernel'], m.params[name + '/bias']
    return ops.dense(x, w.data, b.data, act, alpha, wgrad=w.g(), bgrad=b.g(), anchor=anchor(x.device))


def bn(m, name, x, training, relu=False, out_dtype=torch.float32):
    g, b = m.params[name + '/gamma'], m.params[name + '/beta']
    return ops.batchnorm(x, g.data, b.data, m.params[name + '/moving_mean'].data,
                         m.params[name + '/moving_variance'].data, training, relu,
                         ggrad=g.g(), bgrad=b.g(), anchor=anchor(x.device), out_dtype=out_dtype)


def conv_bn(m, cname, bname, x, training, relu=False, x2=None, upsample=False, y16=True):
    """Conv2D(3x3 'same') -> BatchNormalization [-> ReLU].  Training: the two layers as separate launches (batch statistics
    need the whole convolution output first).  Inference without a tape (`predict`): one launch, BatchNorm folded into
    the convolution epilogue.
    With 16-bit activation storage on (conf.act_storage = 'half', ops.set_activation_storage): the convolution output is stored in the
    16-bit type when the convolution runs on the MFMA fast path (input channels a multiple of 32), the BatchNorm output when the
    caller allows it (`y16`: its consumers read 16-bit tensors) and the channel count is a multiple of 64."""
    w, b = m.params[cname + '/kernel'], m.params.get(cname + '/bias')
    half = ops.act16_dtype()
    Cout = w.shape[3]
    c1 = x.shape[3]
    c2 = x2.shape[3] if x2 is not None else 0
    conv_dt = half if (half is not None and c1 % 32 == 0 and c2 % 32 == 0 and Cout % 64 == 0) else torch.float32
    bn_dt = half if (half is not None and y16 and Cout % 64 == 0) else torch.float32
    if training or torch.is_grad_enabled():
        l = conv(m, cname, x, x2=x2, upsample=upsample, bias_grad=not training, out_dtype=conv_dt)
        return bn(m, bname, l, training, relu=relu, out_dtype=bn_dt)
    return ops.conv2d_bn_infer(x, w.data, b.data if b is not None else None, m.params[bname + '/gamma'].data,
                               m.params[bname + '/beta'].data, m.params[bname + '/moving_mean'].data,
                               m.params[bname + '/moving_variance'].data, relu=relu, x2=x2, upsample=upsample, wkey=(w.uid, w.owner.uid),
                               out_dtype=bn_dt)


def norm_params(m, name, c, norm):
    """parameters of utils/model_utils.normalise(norm) (reference utils/model_utils.py:6-12): BatchNormalization,
    keras_contrib InstanceNormalization() (axis=None: ONE scalar gamma and beta), or the identity Lambda"""
    if norm == 'batch':
        bn_params(m, name, c)
    elif norm == 'instance':
        m.add_param(name + '/gamma', (1,), 'ones')
        m.add_param(name + '/beta', (1,), 'zeros')


def conv_norm(m, cname, nname, x, training, relu=False, x2=None, upsample=False, norm='batch'):
    """Conv2D(3x3 'same') -> normalise(norm) [-> ReLU] (models/unet.py:94-101, utils/model_utils.py:6-22)"""
    if norm == 'batch':
        return conv_bn(m, cname, nname, x, training, relu=relu, x2=x2, upsample=upsample)
    if norm == 'instance':
        # (x - mean) / (std + 1e-3) over (H, W, C) jointly per sample, then the scalar affine and the activation in the FiLM
        # kernel (alpha 0 = ReLU, alpha 1 = linear)
        l = conv(m, cname, x, x2=x2, upsample=upsample)
        g, b = m.params[nname + '/gamma'], m.params[nname + '/beta']
        B, C = l.shape[0], l.shape[3]
        gt = ops.expand_scalar(g.data, B, C, g.g(), anchor(x.device))
        bt = ops.expand_scalar(b.data, B, C, b.g(), anchor(x.device))
        return ops.film(ops.instnorm_spade(l), gt, bt, None, 0.0 if relu else 1.0)
    return conv(m, cname, x, x2=x2, upsample=upsample, act='relu' if relu else None)


# ---- Keras 2.1.6 Adam over the arenas of a set of models -------------------------------------------------------
class Adam(object):
    """keras.optimizers.Adam(lr): beta (0.9, 0.999), epsilon 1e-7, bias correction folded into lr_t.  One instance
    per compiled trainer: the reference gives supervised_trainer and unsupervised_trainer separate states over the
    same weights (models/dafnet.py:155,161)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.iterations = 0
        self.state = {}

    def _lr_t(self):
        t = self.iterations
        return self.lr * math.sqrt(1. - self.beta_2 ** t) / (1. - self.beta_1 ** t)

    def begin_device_step(self, device):
        """graphs.py: advance the iteration count and put this step's lr_t into the device scalar the recorded Adam launches read"""
        self.iterations += 1
        if getattr(self, '_lr_dev', None) is None:
            self._lr_dev = torch.empty(1, dtype=torch.float32, device=device)
        ops.fill_(self._lr_dev, self._lr_t())
        return self._lr_dev

    def step(self, models, lr_dev=None):
        if lr_dev is None:
            self.iterations += 1
        lr_t = self._lr_t() if lr_dev is None else lr_dev
        for m in models:
            st = self.state.get(m.uid)
            if st is None:
                st = (ops.fill_(torch.empty_like(m.arena), 0.0), ops.fill_(torch.empty_like(m.arena), 0.0))
                self.state[m.uid] = st
            ops.adam_step(m.arena, m.grad_arena, st[0], st[1], lr_t, self.beta_1, self.beta_2, self.epsilon, owner=m.uid)


class History(object):
    """keras History: .history[name] is a one-element list.  Values stay on the device until read."""

    def __init__(self):
        self._dev = OrderedDict()

    def record(self, name, value):
        self._dev[name] = value      # later duplicates overwrite earlier ones, like Keras' logs dict

    @property
    def history(self):
        return _LazyHistory(self._dev)


class _LazyHistory(dict):
    def __init__(self, dev):
        super(_LazyHistory, self).__init__()
        self._dev = dev

    def __getitem__(self, k):
        v = self._dev[k]
        return [float(v.item()) if hasattr(v, 'item') else float(v)]

    def __contains__(self, k):
        return k in self._dev

    def keys(self):
        return self._dev.keys()
