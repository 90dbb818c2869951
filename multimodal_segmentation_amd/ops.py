"""Operators of the DAFNet/MMSDNet step: thin autograd nodes around the gfx950 kernels of libmmseg_hip.so.

torch is used for device memory (torch.empty), the stream and the autograd tape that stitches the nodes; every
arithmetic operation on activations, gradients and weights is a kernel of csrc/*.hip reached through
_native.call (C ABI, include/mmseg_hip.h).  Layout: NHWC fp32 (the reference's layout), weights in Keras
layout (conv HWIO, dense [in, out]).
"""
import torch

from . import _native as N
from .parallel import dp

ACT = {None: 0, 'linear': 0, 'relu': 1, 'leaky': 2, 'tanh': 3}
BN_EPS = 1e-3        # keras BatchNormalization default
BN_MOMENTUM = 0.99   # keras BatchNormalization default
IN_EPS = 1e-3        # keras_contrib InstanceNormalization default

_workspaces = {}


def _sid(device):
    """identity of the stream the next launches go to: scratch buffers and cached weight images are per stream, so that independent
    phases of an iteration may run on concurrent streams (conf.multi_stream) without sharing mutable scratch memory"""
    if device.type != 'cuda':
        return 0
    from . import graphs
    st = graphs._recording
    if st is not None and st.mode == 'capture':
        return ('graph', st.uid)       # a recorded step owns its scratch: torch records every graph on one shared capture stream,
                                       # and two recorded steps may later be replayed on concurrent streams
    return torch.cuda.current_stream(device).cuda_stream


def _ws(tag, nfloats, device):
    """Grow-only scratch buffer per (tag, device, stream).  The kernels of one stream run in order, so a buffer can be handed to the
    next kernel as soon as the previous launch has been queued."""
    key = (tag, device, _sid(device))
    buf = _workspaces.get(key)
    n = max(int(nfloats), 1)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1024), dtype=torch.float32, device=device)
        _workspaces[key] = buf
    return buf


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ---- reduced-precision STORAGE of the convolutional trunk (build-defined; BASELINE configs #3 / #5) ------------------------------
_act16 = [None]      # torch.bfloat16 / torch.float16 while 16-bit activation storage is on, else None
_HCODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def set_activation_storage(on):
    """conf.act_storage = 'half': the activations and gradients of the MFMA trunk (UNet conv - BatchNorm - ReLU chains incl. pooling,
    up-sampling and skips; the segmentor's first block) live in HBM in the 16-bit type of the active precision mode.  Master
    weights, weight gradients, statistics, softmax / rounding, losses, Adam and the all-reduce stay fp32.  Needs
    set_conv_precision('bf16' | 'fp16') first.  Returns the previous setting."""
    old = _act16[0]
    if not on:
        _act16[0] = None
        return old is not None
    mode = N.call('mmseg_get_conv_precision')
    if mode == 0:
        raise ValueError("16-bit activation storage needs compute_dtype 'bf16' or 'fp16'")
    _act16[0] = torch.bfloat16 if mode == 1 else torch.float16
    return old is not None


def act16_dtype():
    return _act16[0]


def _h(t):
    """element code of a tensor for the `_t` entry points: 0 fp32, 1 bf16, 2 fp16"""
    return 0 if t is None else _HCODE[t.dtype]


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------------
_PRECISIONS = ['fp32', 'bf16', 'fp16']


def set_conv_precision(mode):
    """'fp32' (default), 'bf16' or 'fp16': precision of the MFMA products of the fast-path convolutions (operands
    rounded to the 16-bit type, fp32 accumulation; tensors in HBM, weight gradients, normalisation, losses and the optimiser
    stay fp32).  Returns the previous mode."""
    old = N.call('mmseg_set_conv_precision', _PRECISIONS.index(mode))
    if _PRECISIONS[old] != mode:
        bump_weight_version()           # the fast-path weight images hold elements of the operand type: cached ones are stale
    return _PRECISIONS[old]


class precision_scope(object):
    """`with precision_scope((compute_dtype, act16)):` -- run the enclosed launches in the precision of ONE model: the kernel library's
    mode (and the storage type of the trunk) is set on entry and put back on exit.  Trainers and `predict` enter the scope of the
    model they belong to, so models of different precisions can live in one process without the caller juggling the library-wide
    switch (round-2 review); `None` leaves whatever is set."""

    def __init__(self, precision):
        self.precision = precision

    def __enter__(self):
        self.prev = None
        if self.precision is not None:
            mode, act16 = self.precision
            cur = (_PRECISIONS[N.call('mmseg_get_conv_precision')], _act16[0] is not None)
            if cur != (mode, bool(act16)):
                self.prev = cur
                set_conv_precision(mode)
                set_activation_storage(bool(act16))
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            set_activation_storage(False)
            set_conv_precision(self.prev[0])
            set_activation_storage(self.prev[1])
        return False


def _conv_geometry(H, W, KH, KW, stride, padding):
    if padding == 'same':
        if stride != 1 or KH % 2 == 0 or KW % 2 == 0:
            raise ValueError("'same' is implemented for stride 1 and odd kernels (all the reference uses)")
        return H, W, KH // 2, KW // 2
    if padding == 'valid':
        return (H - KH) // stride + 1, (W - KW) // stride + 1, 0, 0
    raise ValueError(padding)


_weight_version = [0]
_wprep_cache = {}


_bn_state_version = [0]      # bumped whenever BatchNorm moving statistics may have changed (training-mode forward)
_bnfold_cache = {}


_owner_version = {}          # nn.Model.uid -> version of that model's weights (bumped by its optimiser steps only)


_wprep_batch = {}            # nn.Model.uid -> {'ents': {cache key: (src ptr, image, taps, Cin, Cout, mode, wkey)}, 'table': tensor | None, 'blocks': int}
_wprep_batch_on = [True]     # False: every image is re-laid out lazily by its own launch again (tests compare the two)


def _wprep_register(key, wkey, w, out, ntaps, Cin, Cout, mode):
    """remember a cached mode 0 / 1 image of a model's parameter: the model's next optimiser step refreshes all of them in ONE launch"""
    if not (isinstance(wkey, tuple) and len(wkey) == 2 and isinstance(wkey[0], int) and mode in (0, 1) and w.is_cuda):
        return
    b = _wprep_batch.setdefault(wkey[1], {'ents': {}, 'table': None, 'blocks': 0})
    ent = b['ents'].get(key)
    if ent is None or ent[0] != w.data_ptr() or ent[1] is not out:
        b['ents'][key] = (w.data_ptr(), out, ntaps, Cin, Cout, mode, wkey)
        b['table'] = None


def _wprep_refresh(owner):
    """all registered images of `owner`'s parameters in one launch (mmseg_conv2d_wprep_batch), stamped with the current version"""
    b = _wprep_batch.get(owner)
    if not b or not b['ents'] or not _wprep_batch_on[0]:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    ents = list(b['ents'].items())
    dev = ents[0][1][1].device
    if b['table'] is None:
        rows, blk = [], 0
        for _, (src, out, ntaps, Cin, Cout, mode, _wk) in ents:
            rows.append([src, out.data_ptr(), ntaps, Cin, Cout, mode, blk, 0])
            blk += ntaps * ((Cin + 31) // 32) * ((Cout + 31) // 32)
        rows.append([0, 0, 0, 0, 0, 0, blk, 0])
        b['table'] = torch.tensor(rows, dtype=torch.int64).to(dev)
        b['blocks'] = blk
    N.call('mmseg_conv2d_wprep_batch', b['table'], len(ents), b['blocks'])
    for key, ent in ents:
        _wprep_cache[key] = (_wver(ent[6]), ent[1])


def bump_weight_version(owner=None):
    """Invalidate the cached weight re-layouts (called whenever weights change: optimiser step, set_weights).  `owner` = id of the
    nn.Model whose arena changed: only the images of ITS parameters become stale -- a discriminator's Adam step no longer makes the
    generator's ~100 forward images stale (they were re-laid out three times per iteration: after the generator step and again
    after the mask- and image-discriminator steps).  Without `owner` everything is invalidated (set_weights, precision switch,
    graph capture, broadcast)."""
    if owner is None:
        _weight_version[0] += 1
        _wprep_batch.clear()         # arenas may have moved (set_weights, broadcast): the images re-register as they are re-laid out
    else:
        _owner_version[owner] = _owner_version.get(owner, 0) + 1
        _wprep_refresh(owner)
    _bn_state_version[0] += 1


def _key_owner(key):
    """Model.uid a cache key belongs to, or None (keys start with the wkey = (Param.uid | ('pair', Param.uid), Model.uid))"""
    wk = key[0]
    return wk[1] if isinstance(wk, tuple) and len(wk) == 2 else None


def evict_owner(owner):
    """Drop every cached weight image / fused operand / BatchNorm fold of the model with this uid (nn.Model registers this as its
    finaliser): a process that builds many models does not keep the images of the dead ones in HBM."""
    for cache in (_wprep_cache, _pair_cache, _bnfold_cache):
        for k in [k for k in cache if _key_owner(k) == owner]:
            del cache[k]
    _wprep_batch.pop(owner, None)
    _owner_version.pop(owner, None)


def release_caches(workspaces=True):
    """Forget every cached weight image, fused operand and BatchNorm fold (they are rebuilt on demand) and, optionally, the
    grow-only scratch buffers of all streams.  Must not be called while a recorded hipGraph that owns scratch is alive."""
    _wprep_cache.clear()
    _pair_cache.clear()
    _bnfold_cache.clear()
    _wprep_batch.clear()
    if workspaces:
        _workspaces.clear()


def release_graph_workspaces(uid):
    """the scratch buffers a recorded step owned (graphs.FitGraph registers this as its finaliser)"""
    sid = ('graph', uid)
    for k in [k for k in _workspaces if k[2] == sid]:
        del _workspaces[k]
    for cache in (_wprep_cache, _pair_cache, _bnfold_cache):       # ... and the weight images written inside the recording
        for k in [k for k in cache if k[-1] == sid]:
            del cache[k]
    for b in _wprep_batch.values():
        stale = [k for k in b['ents'] if k[-1] == sid]
        for k in stale:
            del b['ents'][k]
        if stale:
            b['table'] = None


def cache_footprint():
    """(entries, bytes) held by the caches and the scratch buffers -- for leak checks"""
    n = b = 0
    for cache in (_wprep_cache, _pair_cache, _bnfold_cache):
        for ent in cache.values():
            for t in ent[1:]:
                if isinstance(t, torch.Tensor):
                    n, b = n + 1, b + t.numel() * t.element_size()
    for t in _workspaces.values():
        n, b = n + 1, b + t.numel() * t.element_size()
    return n, b


def _wver(wkey):
    """version stamp of a cached image: (global version, version of the owning model); wkey = (Param.uid, Model.uid) -- serial
    numbers that are never reused (object ids are recycled by the interpreter, device addresses by the caching allocator)"""
    if isinstance(wkey, tuple):
        return (_weight_version[0], _owner_version.get(wkey[1], 0))
    return (_weight_version[0], 0)


def _wprep(w, KH, KW, Cin, Cout, mode, wkey=None):
    """[N][K] re-layout of a conv kernel for the fast path (mode 0: forward, 1: data gradient).  Cached per weight
    version when the caller identifies the weight (`wkey`, a parameter of an nn.Model whose arena is stable and whose
    every update bumps the version); anonymous weights are re-laid out on every call."""
    if wkey is None:
        out = _ws('wprep%d' % mode, w.numel(), w.device)[:w.numel()]
        N.call('mmseg_conv2d_wprep', w, out, KH, KW, Cin, Cout, mode)
        return out
    key = (wkey, w.data_ptr(), mode, _sid(w.device))
    ent = _wprep_cache.get(key)
    if ent is not None and ent[0] == _wver(wkey) and ent[1].numel() == w.numel():
        return ent[1]
    out = ent[1] if (ent is not None and ent[1].numel() == w.numel()) else torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    N.call('mmseg_conv2d_wprep', w, out, KH, KW, Cin, Cout, mode)
    _wprep_cache[key] = (_wver(wkey), out)
    _wprep_register(key, wkey, w, out, KH * KW, Cin, Cout, mode)
    return out


def _wprep_col8(w, Cout, wkey=None):
    """fast-path weight image of a 3x3 kernel with 8 input channels read as a 1x1 kernel over the 96-column im2col rows of
    mmseg_im2col8_t: [72][Cout] + 24 zero rows -> [Cout][96]; cached per weight version like _wprep"""
    key = (wkey, w.data_ptr(), 'col8', _sid(w.device))
    ent = _wprep_cache.get(key) if wkey is not None else None
    if ent is not None and ent[0] == _wver(wkey):
        return ent[1]
    if ent is not None:
        pad, out = ent[2], ent[1]
    else:
        pad = torch.zeros(96 * Cout, dtype=torch.float32, device=w.device)
        out = torch.empty(96 * Cout, dtype=torch.float32, device=w.device)
    pad[:72 * Cout].copy_(w.reshape(-1))
    N.call('mmseg_conv2d_wprep', pad, out, 1, 1, 96, Cout, 0)
    if wkey is not None:
        _wprep_cache[key] = (_wver(wkey), out, pad)
    return out


def _wprep_parity_all(w, KH, KW, Cin, Cout, stride, taps, wkey=None):
    """The sub-kernels of all stride x stride parity classes back to back in (ph, pw) raster order (the operand of
    mmseg_conv2d_dgrad_parity_all); cached per weight version like _wprep."""
    sizes = [taps[0][qh] * taps[1][qw] * Cin * Cout for qh in range(stride) for qw in range(stride)]
    n = sum(sizes)
    key = (wkey, w.data_ptr(), 'parity_all', stride, _sid(w.device))
    ent = _wprep_cache.get(key) if wkey is not None else None
    if ent is not None and ent[0] == _wver(wkey) and ent[1].numel() == n:
        return ent[1]
    if wkey is None:
        out = _ws('wprep_parity_all', n, w.device)[:n]
    else:
        out = ent[1] if (ent is not None and ent[1].numel() == n) else torch.empty(n, dtype=torch.float32, device=w.device)
    N.call('mmseg_conv2d_wprep_parity_all', w, out, KH, KW, Cin, Cout, stride)
    if wkey is not None:
        _wprep_cache[key] = (_wver(wkey), out)
    return out


def _conv_fwd_raw(x1, x2, w, wt, bias, y, y2, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, transposed,
                  act, alpha, nsplit1):
    # host-side shape checks: the kernel trusts these numbers
    assert x1.numel() == B * (H >> ups) * (W >> ups) * C1, 'x1 shape/geometry mismatch'
    assert (x2 is None and C2 == 0) or x2.numel() == B * H * W * C2, 'x2 shape/geometry mismatch'
    assert (w if w is not None else wt).numel() == KH * KW * (C1 + C2) * Cout, 'kernel shape mismatch'
    assert bias is None or bias.numel() == Cout
    if y2 is None:
        assert y.numel() == B * Ho * Wo * Cout
    else:
        assert y.numel() == B * Ho * Wo * nsplit1 and y2.numel() == B * Ho * Wo * (Cout - nsplit1)
    io = (1 if _h(x1) else 0) | (2 if _h(x2) else 0) | (4 if _h(y) else 0)
    if io:      # 16-bit tensors in HBM (reduced-precision storage)
        assert y2 is None or _h(y2) == _h(y)
        N.call('mmseg_conv2d_fwd_t', x1, x2, w, wt, bias, y, y2, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw,
               ups, transposed, act, float(alpha), nsplit1, io)
        return
    N.call('mmseg_conv2d_fwd', x1, x2, w, wt, bias, y, y2, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw,
           ups, transposed, act, float(alpha), nsplit1)


def _grad_done(*grads):
    """Tell the data-parallel tracker that the launches accumulating into these gradient-arena views have been queued
    (parallel/dp.py: the arena's all-reduce starts when its last accumulation is in the stream)."""
    tr = dp.current_tracker()
    if tr is None:
        return
    for t in grads:
        owner = getattr(t, '_owner', None) if t is not None else None
        if owner is not None:
            tr.done(owner, getattr(t, '_seg', 0))


def _accumulate(dst, src):
    """dst += src (weight-gradient accumulation into the gradient arena)."""
    N.call('mmseg_axpby', dst, src, dst, dst.numel(), 1.0, 1.0)


def _wgrad_launch(x1, x2, g, dw_flat, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, accumulate):
    """dW (+)= weight gradient of a convolution; x1 / x2 / g as they are stored (fp32 or the 16-bit storage type)"""
    need = N.call('mmseg_conv2d_wgrad_workspace', B, Ho, Wo, C1 + C2, Cout, KH, KW)
    ws = _ws('wgrad', need, g.device)
    if (_h(x1) or _h(x2) or _h(g)) and N.call('mmseg_conv2d_wgrad_t_supported', Ho, Wo, stride, C1, C2, Cout):
        # 16-bit operands in HBM: the transposed-staging kernel reads them as they are
        assert x2 is None or _h(x2) == _h(x1)
        N.call('mmseg_conv2d_wgrad_t', x1, x2, g, dw_flat, ws, ws.numel(), B, H, W, C1, C2, Ho, Wo, Cout, KH, KW,
               stride, ph, pw, ups, accumulate, (1 if _h(x1) else 0) | (4 if _h(g) else 0))
    elif _h(x1) or _h(x2) or _h(g):
        # geometry the 16-bit kernel does not take (tiny planes): widen the operands (a copy, no arithmetic)
        f = lambda t: t if t is None or _h(t) == 0 else t.float()
        N.call('mmseg_conv2d_wgrad', f(x1), f(x2), f(g), dw_flat, ws, ws.numel(), B, H, W, C1, C2, Ho, Wo, Cout, KH, KW,
               stride, ph, pw, ups, accumulate)
    else:
        N.call('mmseg_conv2d_wgrad', x1, x2, g, dw_flat, ws, ws.numel(), B, H, W, C1, C2, Ho, Wo, Cout, KH, KW,
               stride, ph, pw, ups, accumulate)


class _Conv2d(torch.autograd.Function):
    """inputs: activations x1 [, x2], the tape anchor, then non-differentiable arguments.  Weight / bias gradients
    are accumulated into `wgrad` / `bgrad` (views of the owner's gradient arena) instead of being returned."""

    @staticmethod
    def forward(ctx, x1, x2, anchor, w, bias, stride, padding, act, alpha, ups, wgrad, bgrad, wkey, out_dtype=torch.float32):
        x1 = _c(x1)
        x2 = _c(x2) if x2 is not None else None
        B, H1, W1, C1 = x1.shape
        H, W = (H1 * 2, W1 * 2) if ups else (H1, W1)
        C2 = 0 if x2 is None else x2.shape[3]
        if x2 is not None:
            assert x2.shape[:3] == (B, H, W)
        KH, KW, Cin, Cout = w.shape
        assert Cin == C1 + C2, 'kernel expects %d input channels, got %d' % (Cin, C1 + C2)
        Ho, Wo, ph, pw = _conv_geometry(H, W, KH, KW, stride, padding)
        y = _new((B, Ho, Wo, Cout), x1, out_dtype)
        prec = N.call('mmseg_get_conv_precision')
        if prec and C1 == 8 and C2 == 0 and KH == 3 and KW == 3 and stride == 1 and not ups and (Ho, Wo, ph, pw) == (H, W, 1, 1) \
                and Cout % 64 == 0 and _h(x1) in (0, prec) and B * H * W * 96 * 2 < (1 << 31) - 64:
            # reduced-precision modes, 8 input channels (the SPADE units' shared convolution, the segmentor's first): K = 72 is three
            # gathers of the generic kernel per output tile; instead the 72 (+24 zero) operand columns of every pixel are written
            # once as 16-bit rows and the product runs as a 1x1 convolution on the 16-bit MFMA fast path
            if W % 32 == 0 and _h(y) in (0, prec) and ACT[act] <= 2 and N.call('mmseg_conv16_mode', -1) != 0:
                # round 4: one launch, no im2col tensor -- a lane's MFMA operand is one pixel's 8 channels of one tap, read as it lies;
                # the weights stay in registers (conv8h_kernel, csrc/conv16.hpp)
                assert w.numel() == 72 * Cout and (bias is None or bias.numel() == Cout) and y.numel() == B * H * W * Cout
                N.call('mmseg_conv8h_fwd_t', x1, w, bias, y, B, H, W, Cout, ACT[act], float(alpha), _h(x1), _h(y))
            else:
                half = torch.bfloat16 if prec == 1 else torch.float16
                xcol = torch.empty((B, H, W, 96), dtype=half, device=x1.device)
                N.call('mmseg_im2col8_t', x1, xcol, B, H, W, _h(x1), prec)
                _conv_fwd_raw(xcol, None, None, _wprep_col8(w, Cout, wkey), bias, y, None, B, H, W, 96, 0, Ho, Wo, Cout, 1, 1, 1, 0, 0, 0, 0,
                              ACT[act], alpha, 0)
        else:
            wt = _wprep(w, KH, KW, Cin, Cout, 0, wkey) if N.call('mmseg_conv2d_fast_path', C1, C2, Cout, 0) else None
            _conv_fwd_raw(x1, x2, w, wt, bias, y, None, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, int(ups), 0,
                          ACT[act], alpha, 0)
        ctx.geom = (B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, int(ups), ACT[act], alpha)
        ctx.wgrad, ctx.bgrad, ctx.wkey = wgrad, bgrad, wkey
        ctx.w = w     # plain (non-leaf) weight view: not tracked by autograd
        ctx.save_for_backward(x1, x2, y if ACT[act] else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, y = ctx.saved_tensors
        w = ctx.w
        B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, act, alpha = ctx.geom
        dy = _c(dy)
        M = B * Ho * Wo
        bias_done = False
        if act:
            assert _h(dy) == _h(y), 'the gradient of a tensor is stored like the tensor'
            g = _new(dy.shape, dy, dy.dtype)      # act16_bwd_kernel stores dx with dy's element type
            if ctx.bgrad is not None and Cout % 64 == 0:
                # activation gradient and bias gradient (column sums of g) in one pass over dy / y
                ws = _ws('colsum', N.call('mmseg_colsum_workspace_floats', M, Cout), dy.device)
                N.call('mmseg_act_bwd_bias_t', dy, y, g, ctx.bgrad, ws, M, Cout, act, float(alpha), 1, _h(dy))
                bias_done = True
            elif _h(dy):
                N.call('mmseg_act_bwd_bias_t', dy, y, g, None, None, M, Cout, act, float(alpha), 0, _h(dy))
            else:
                N.call('mmseg_act_bwd', dy, y, g, dy.numel(), act, float(alpha))
        else:
            g = dy
        need_x1, need_x2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx1 = dx2 = None
        if ctx.bgrad is not None and not bias_done:
            ws = _ws('colsum', N.call('mmseg_colsum_workspace_floats', M, Cout), dy.device)
            N.call('mmseg_colsum', g if _h(g) == 0 else g.float(), ctx.bgrad, ws, M, Cout, 1.0, 1)
        if ctx.wgrad is not None:
            # accumulates straight into the gradient-arena view (the final slab reduction adds to it)
            _wgrad_launch(x1, x2, g, ctx.wgrad.view(-1), B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, 1)
        _grad_done(ctx.wgrad, ctx.bgrad)
        if need_x1 or (x2 is not None and need_x2):
            Cin = C1 + C2
            tr = 1 if stride > 1 else 0
            d1 = _new((B, H, W, C1), dy, x1.dtype)          # a gradient is stored like its tensor
            d2 = _new((B, H, W, C2), dy, x2.dtype) if C2 else None
            assert d2 is None or d2.dtype == d1.dtype
            taps = [[N.call('mmseg_conv2d_parity_taps', k, stride, q) for q in range(stride)] for k in (KH, KW)] if tr else None
            if tr and stride == 2 and KH == 4 and KW == 4 and ph == 0 and pw == 0 and C2 == 0 and not ups and Cout == 64 and \
                    Cin in (1, 4) and N.call('mmseg_conv2d_fast_path', 64, 0, 64, 0):
                # first layer of a discriminator: N = Cin GEMM -> direct kernel
                N.call('mmseg_conv2d_dgrad_s2k4_smallc', g, w, d1, B, H, W, Cin, Ho, Wo, Cout)
            elif stride == 1 and C2 == 0 and not ups and Cin % 4 == 0 and Cin <= 16 and KH * KW > 1 and KH * KW * Cin <= 256 and \
                    not (Cin == 8 and Cout == 8) and N.call('mmseg_conv2d_fast_path', Cout, 0, KH * KW * Cin, 0):
                # few input channels: ONE 1x1 GEMM of the gradient against all taps (N = taps * Cin instead of a 32-wide tile
                # that is mostly padding, dy read once instead of once per tap), then the shifted planes are summed
                nt = KH * KW * Cin
                T = _ws('dgrad_taps', B * Ho * Wo * nt, dy.device)[:B * Ho * Wo * nt]
                # the Keras kernel [KH, KW, Cin, Cout] read as [taps * Cin][Cout] IS the fast layout of that 1x1 convolution
                # (in the reduced-precision modes the image holds 16-bit elements: converted copy, cached like the other images)
                wimg = w.reshape(-1) if N.call('mmseg_get_conv_precision') == 0 else _wprep(w, KH, KW, Cin, Cout, 2, ctx.wkey)
                _conv_fwd_raw(g, None, None, wimg, None, T, None, B, Ho, Wo, Cout, 0, Ho, Wo, nt, 1, 1, 1, 0, 0, 0, 0, 0, 0.0, 0)
                N.call('mmseg_conv2d_dgrad_tapsum', T, d1, B, H, W, Ho, Wo, Cin, KH, KW, ph, pw)
            elif tr and stride == 2 and C2 == 0 and not ups and N.call('mmseg_conv2d_fast_path', Cout, 0, Cin, 0) and \
                    min(taps[0] + taps[1]) > 0:
                # strided convolution: the parity classes of the input pixels, each an exact stride-1 convolution, batched
                # into one launch (a single class does not fill the chip)
                wp = _wprep_parity_all(w, KH, KW, Cin, Cout, stride, taps, ctx.wkey)
                N.call('mmseg_conv2d_dgrad_parity_all', g, wp, d1, B, Ho, Wo, Cout, H, W, Cin, KH, KW, stride)
            else:
                if N.call('mmseg_conv2d_fast_path', Cout, 0, Cin, tr):
                    wf, wt = None, _wprep(w, KH, KW, Cin, Cout, 1, ctx.wkey)     # [Cin][flipped taps][Cout]
                else:
                    wf, wt = _ws('wflip', w.numel(), dy.device)[:w.numel()], None
                    N.call('mmseg_conv2d_wflip', w, wf, KH, KW, Cin, Cout)
                # data gradient = convolution of g with the flipped kernel; fractionally strided when stride > 1
                _conv_fwd_raw(g, None, wf, wt, None, d1, d2, B, Ho, Wo, Cout, 0, H, W, Cin, KH, KW, stride, KH - 1 - ph,
                              KW - 1 - pw, 0, tr, 0, 0.0, C1 if C2 else 0)
            if ups:
                dx1 = _new((B, H // 2, W // 2, C1), dy, d1.dtype)
                if _h(d1):
                    N.call('mmseg_upsample2_bwd_t', d1, dx1, B, H // 2, W // 2, C1, _h(d1))
                else:
                    N.call('mmseg_upsample2_bwd', d1, dx1, B, H // 2, W // 2, C1)
            else:
                dx1 = d1
            dx2 = d2
        return (dx1, dx2) + (None,) * 12


def conv2d_bn_infer(x, w, cbias, gamma, beta, mov_mean, mov_var, relu=False, x2=None, upsample=False, wkey=None,
                    out_dtype=torch.float32):
    """`predict` of Conv2D -> BatchNormalization [-> ReLU] as ONE launch: the moving statistics are folded into a
    per-channel scale and bias of the convolution epilogue (no tape: inference only)."""
    x1 = _c(x)
    x2 = _c(x2) if x2 is not None else None
    B, H1, W1, C1 = x1.shape
    H, W = (2 * H1, 2 * W1) if upsample else (H1, W1)
    C2 = x2.shape[3] if x2 is not None else 0
    KH, KW, Cin, Cout = w.shape
    assert Cin == C1 + C2
    Ho, Wo, ph, pw = _conv_geometry(H, W, KH, KW, 1, 'same')
    # folded scale / bias: cached per layer until the weights or the moving statistics change (a pool evaluates every
    # encoder layer four times per iteration with the same parameters)
    key = (wkey, gamma.data_ptr(), 'bnfold', _sid(gamma.device))
    ent = _bnfold_cache.get(key) if wkey is not None else None
    if ent is not None and ent[0] == (_wver(wkey), _bn_state_version[0]):
        ss = ent[1]
    else:
        ss = ent[1] if ent is not None else _new((2, Cout), x1)
        N.call('mmseg_bn_infer_fold', gamma, beta, mov_mean, mov_var, cbias, ss[0], ss[1], Cout, BN_EPS)
        if wkey is not None:
            _bnfold_cache[key] = ((_wver(wkey), _bn_state_version[0]), ss)
    y = _new((B, Ho, Wo, Cout), x1, out_dtype)
    wt = _wprep(w, KH, KW, Cin, Cout, 0, wkey) if N.call('mmseg_conv2d_fast_path', C1, C2, Cout, 0) else None
    io = (1 if _h(x1) else 0) | (2 if _h(x2) else 0) | (4 if _h(y) else 0)
    if io:
        N.call('mmseg_conv2d_fwd_scaled_t', x1, x2, w, wt, ss[1], ss[0], y, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, 1, ph, pw,
               int(bool(upsample)), ACT['relu' if relu else None], 0.0, io)
    else:
        N.call('mmseg_conv2d_fwd_scaled', x1, x2, w, wt, ss[1], ss[0], y, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, 1, ph, pw,
               int(bool(upsample)), ACT['relu' if relu else None], 0.0)
    return y


def conv2d(x, w, bias=None, stride=1, padding='same', act=None, alpha=0.0, x2=None, upsample=False,
           wgrad=None, bgrad=None, anchor=None, wkey=None, out_dtype=torch.float32):
    """keras Conv2D on NHWC (+ fused nearest x2 up-sampling of x, + fused channel concat with x2, + fused
    bias/activation epilogue).  `wgrad`/`bgrad`: gradient-arena views to accumulate into (None = frozen)."""
    if upsample and (x.shape[3] % 4 != 0):
        raise ValueError('fused up-sampling needs C % 4 == 0')
    if wgrad is None and bgrad is None:
        anchor = None
    return _Conv2d.apply(x, x2, anchor, w, bias, stride, padding, act, alpha, bool(upsample), wgrad, bgrad, wkey, out_dtype)


# ------------------------------------------------------------------------------------------------------
# batch norm (+ReLU)
# ------------------------------------------------------------------------------------------------------
class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, gamma, beta, mov_mean, mov_var, relu, ggrad, bgrad, out_dtype=torch.float32):
        _bn_state_version[0] += 1          # the moving statistics are about to change: folded inference parameters are stale
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        stats = _new((4, C), x)  # mean, invstd, scale, shift
        ws = _ws('norm', N.call('mmseg_norm_workspace_floats', C), x.device)
        ctx.sync = dp.sync_bn()
        hx, hy = _h(x), _HCODE[out_dtype]
        ctx.h = (hx, hy)
        if hx or hy:       # 16-bit storage of the input and / or the output (csrc/act16.hip): same arithmetic, 8-byte accesses
            assert not ctx.sync, 'sync_bn and 16-bit activation storage are not combined'
            N.call('mmseg_bn_stats_t', x, gamma, beta, stats[0], stats[1], stats[2], stats[3], mov_mean, mov_var, ws, M, C,
                   BN_EPS, BN_MOMENTUM, hx)
            y = _new(x.shape, x, out_dtype)
            N.call('mmseg_bn_apply_t', x, stats[2], stats[3], y, M, C, int(relu), hx, hy)
            ctx.relu = bool(relu)
            ctx.gamma, ctx.ggrad, ctx.bgrad = gamma, ggrad, bgrad
            ctx.save_for_backward(x, y if relu else None, stats)
            return y
        if ctx.sync:
            # statistics over the global batch: (mean, biased variance) of this rank's rows, gathered in rank order, combined
            local = _new((2, C), x)
            N.call('mmseg_bn_stats_local', x, local, ws, M, C)
            gathered = dp.all_gather_rows(local.reshape(-1))
            N.call('mmseg_bn_stats_combine', gathered, dp.world_size(), gamma, beta, stats[0], stats[1], stats[2], stats[3],
                   mov_mean, mov_var, M * dp.world_size(), C, BN_EPS, BN_MOMENTUM)
        else:
            N.call('mmseg_bn_stats', x, gamma, beta, stats[0], stats[1], stats[2], stats[3], mov_mean, mov_var, ws, M, C,
                   BN_EPS, BN_MOMENTUM)
        y = _new(x.shape, x)
        N.call('mmseg_bn_apply', x, stats[2], stats[3], y, M, C, int(relu))
        ctx.relu = bool(relu)
        ctx.gamma, ctx.ggrad, ctx.bgrad = gamma, ggrad, bgrad
        ctx.save_for_backward(x, y if (relu and ctx.sync) else None, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats = ctx.saved_tensors
        dy = _c(dy)
        C = x.shape[-1]
        M = x.numel() // C
        dx = _new(x.shape, x, x.dtype)
        coef = _ws('bn_coef', 3 * C, x.device)
        ws = _ws('norm', N.call('mmseg_norm_workspace_floats', C), x.device)
        if ctx.h != (0, 0):
            assert (ctx.ggrad is None) == (ctx.bgrad is None), 'BatchNorm gamma and beta are trained or frozen together'
            N.call('mmseg_bn_bwd_t', dy, y, x, ctx.gamma, stats[0], stats[1], dx, ctx.ggrad, ctx.bgrad, coef, ws, M, C, int(ctx.relu), 1,
                   ctx.h[0], ctx.h[1])
            _grad_done(ctx.ggrad, ctx.bgrad)
            return (dx,) + (None,) * 9
        if ctx.sync:
            # sums of this rank -> dgamma / dbeta (averaged over ranks with the other weight gradients); sums over all ranks -> dx
            local = _new((2, C), x)
            N.call('mmseg_bn_bwd_sums', dy, y, x, stats[0], stats[1], local, ws, M, C, int(ctx.relu))
            glob = dp.all_reduce_sum(local.clone())
            if ctx.ggrad is not None and ctx.bgrad is not None:
                N.call('mmseg_bn_bwd_finish', local, glob, ctx.gamma, stats[0], stats[1], ctx.ggrad, ctx.bgrad, coef, C,
                       M * dp.world_size(), 1)
            else:
                assert ctx.ggrad is None and ctx.bgrad is None, 'BatchNorm gamma and beta are trained or frozen together'
                N.call('mmseg_bn_bwd_finish', local, glob, ctx.gamma, stats[0], stats[1], None, None, coef, C, M * dp.world_size(), 0)
            N.call('mmseg_bn_bwd_apply', dy, y, x, coef, dx, M, C, int(ctx.relu))
            _grad_done(ctx.ggrad, ctx.bgrad)
            return (dx,) + (None,) * 9
        # (the ReLU mask is recomputed from x with the forward pass's scale / shift: the saved output is not read again)
        if ctx.ggrad is not None and ctx.bgrad is not None:
            # the final reduction adds dgamma / dbeta straight into the gradient-arena views
            N.call('mmseg_bn_bwd_x', dy, x, stats[2], stats[3], ctx.gamma, stats[0], stats[1], dx, ctx.ggrad, ctx.bgrad, coef, ws, M, C,
                   int(ctx.relu), 1)
        else:
            tmp = _ws('bn_dgb', 2 * C, x.device)
            dgamma, dbeta = tmp[:C], tmp[C:2 * C]
            N.call('mmseg_bn_bwd_x', dy, x, stats[2], stats[3], ctx.gamma, stats[0], stats[1], dx, dgamma, dbeta, coef, ws, M, C,
                   int(ctx.relu), 0)
            if ctx.ggrad is not None:
                _accumulate(ctx.ggrad, dgamma)
            if ctx.bgrad is not None:
                _accumulate(ctx.bgrad, dbeta)
        _grad_done(ctx.ggrad, ctx.bgrad)
        return (dx,) + (None,) * 9


def batchnorm(x, gamma, beta, mov_mean, mov_var, training, relu=False, ggrad=None, bgrad=None, anchor=None,
              out_dtype=torch.float32):
    """keras BatchNormalization(axis=-1) [+ ReLU].  training: batch statistics and in-place moving-average update
    (what `fit` does); otherwise the moving statistics (what `predict` does)."""
    if training:
        if ggrad is None and bgrad is None:
            anchor = None
        return _BatchNormTrain.apply(x, anchor, gamma, beta, mov_mean, mov_var, relu, ggrad, bgrad, out_dtype)
    x = _c(x)
    C = x.shape[-1]
    M = x.numel() // C
    ss = _new((2, C), x)
    N.call('mmseg_bn_infer_prep', gamma, beta, mov_mean, mov_var, ss[0], ss[1], C, BN_EPS)
    y = _new(x.shape, x, out_dtype)
    if _h(x) or _h(y):
        N.call('mmseg_bn_apply_t', x, ss[0], ss[1], y, M, C, int(relu), _h(x), _h(y))
    else:
        N.call('mmseg_bn_apply', x, ss[0], ss[1], y, M, C, int(relu))
    return y


# ------------------------------------------------------------------------------------------------------
# pooling / softmax / rounding
# ------------------------------------------------------------------------------------------------------
class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, H, W, C = x.shape
        y = _new((B, H // 2, W // 2, C), x, x.dtype)
        if _h(x):
            N.call('mmseg_maxpool2_fwd_t', x, y, B, H, W, C, _h(x))
        else:
            N.call('mmseg_maxpool2_fwd', x, y, B, H, W, C)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = _new(x.shape, x, x.dtype)
        if _h(x):
            N.call('mmseg_maxpool2_bwd_t', x, y, _c(dy), dx, B, H, W, C, _h(x))
        else:
            N.call('mmseg_maxpool2_bwd', x, y, _c(dy), dx, B, H, W, C)
        return dx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _MaxPool2Skip(torch.autograd.Function):
    """-> (MaxPooling2D(2)(x), x): a down-path activation of the UNet feeds the pooling AND the skip Concatenate (reference
    models/unet.py:39-51,69-84).  Handing out the second use from the same node lets the backward pass add the two gradients inside
    the pooling-gradient kernel (one pass) instead of the autograd engine adding them with a torch kernel afterwards."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        x = _c(x)
        B, H, W, C = x.shape
        y = _new((B, H // 2, W // 2, C), x, x.dtype)
        if _h(x):
            N.call('mmseg_maxpool2_fwd_t', x, y, B, H, W, C, _h(x))
        else:
            N.call('mmseg_maxpool2_fwd', x, y, B, H, W, C)
        ctx.save_for_backward(x, y)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        if dy is None:
            return dskip
        x, y = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = _new(x.shape, x, x.dtype)
        add = _c(dskip) if dskip is not None else None
        assert add is None or add.dtype == x.dtype, 'the gradient of a tensor is stored like the tensor'
        N.call('mmseg_maxpool2_bwd_add_t', x, y, _c(dy), add, dx, B, H, W, C, _h(x))
        return dx


def maxpool2_skip(x):
    """-> (pooled, skip): `skip` is x for its second consumer (see _MaxPool2Skip)"""
    return _MaxPool2Skip.apply(x)


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, H, W, C = x.shape
        y = _new((B, 2 * H, 2 * W, C), x)
        N.call('mmseg_upsample2_fwd', x, y, B, H, W, C)
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dx = _new(ctx.shape, dy)
        N.call('mmseg_upsample2_bwd', _c(dy), dx, B, H, W, C)
        return dx


def upsample2(x):
    """keras UpSampling2D(size=2) (nearest)."""
    return _Upsample2.apply(x)


class _Subsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f):
        x = _c(x)
        B, H, W, C = x.shape
        assert H % f == 0 and W % f == 0
        y = _new((B, H // f, W // f, C), x)
        N.call('mmseg_subsample_fwd', x, y, B, H // f, W // f, C, f)
        ctx.meta = (B, H, W, C, f)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, f = ctx.meta
        dx = _new((B, H, W, C), dy)
        N.call('mmseg_subsample_bwd', _c(dy), dx, B, H // f, W // f, C, f)
        return dx, None


def resize_nearest_down(x, Ho, Wo):
    """tf.image.resize_nearest_neighbor(x, [Ho, Wo]) for integer down-sampling factors (src = dst * f)."""
    B, H, W, C = x.shape
    if (H, W) == (Ho, Wo):
        return x
    if H % Ho or W % Wo or H // Ho != W // Wo:
        raise NotImplementedError('nearest resize: only integer down-sampling factors (%dx%d -> %dx%d)' % (H, W, Ho, Wo))
    return _Subsample.apply(x, H // Ho)


class _SoftmaxRound(torch.autograd.Function):
    """-> (p, s): channel softmax and its half-to-even rounding with the straight-through gradient of
    layers/rounding.py:40-42 (the gradient reaching s is passed to p unchanged)."""

    @staticmethod
    def forward(ctx, x, want_round):
        x = _c(x)
        C = x.shape[-1]
        p = _new(x.shape, x)
        s = _new(x.shape, x) if want_round else None
        N.call('mmseg_softmax_fwd', x, p, s, x.numel() // C, C)
        ctx.save_for_backward(p)
        return p, s

    @staticmethod
    def backward(ctx, dp, ds):
        (p,) = ctx.saved_tensors
        C = p.shape[-1]
        if dp is None:
            g = _c(ds)
        elif ds is None:
            g = _c(dp)
        else:
            g = _new(p.shape, p)
            N.call('mmseg_axpby', _c(dp), _c(ds), g, p.numel(), 1.0, 1.0)
        dx = _new(p.shape, p)
        N.call('mmseg_softmax_bwd', g, p, dx, p.numel() // C, C)
        return dx, None


def softmax(x):
    return _SoftmaxRound.apply(x, False)[0]


def softmax_round(x):
    """-> (softmax, rounded softmax)"""
    return _SoftmaxRound.apply(x, True)


# ------------------------------------------------------------------------------------------------------
# dense
# ------------------------------------------------------------------------------------------------------
_DENSE_ROWS = 32


class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, w, bias, act, alpha, wgrad, bgrad):
        x = _c(x)
        R, K = x.shape
        K2, Nn = w.shape
        assert K == K2, 'dense: input has %d features, kernel expects %d' % (K, K2)
        y = _new((R, Nn), x)
        for r0 in range(0, R, _DENSE_ROWS):          # the kernels keep <= 32 rows in registers: larger batches go in row groups
            r = min(_DENSE_ROWS, R - r0)
            ws = _ws('dense', N.call('mmseg_dense_workspace_floats', r, K, Nn), x.device)
            N.call('mmseg_dense_fwd', x[r0:r0 + r], w, bias, y[r0:r0 + r], ws, r, K, Nn, ACT[act], float(alpha))
        ctx.act, ctx.alpha = ACT[act], alpha
        ctx.w, ctx.wgrad, ctx.bgrad = w, wgrad, bgrad
        ctx.save_for_backward(x, y if ACT[act] else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        w = ctx.w
        R, K = x.shape
        Nn = w.shape[1]
        dy = _c(dy)
        if ctx.act:
            g = _new(dy.shape, dy)
            N.call('mmseg_act_bwd', dy, y, g, dy.numel(), ctx.act, float(ctx.alpha))
        else:
            g = dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new(x.shape, x)
            for r0 in range(0, R, _DENSE_ROWS):
                r = min(_DENSE_ROWS, R - r0)
                N.call('mmseg_dense_dgrad', g[r0:r0 + r], w, dx[r0:r0 + r], r, K, Nn)
        if ctx.wgrad is not None:
            for r0 in range(0, R, _DENSE_ROWS):          # accumulates straight into the gradient-arena view
                r = min(_DENSE_ROWS, R - r0)
                N.call('mmseg_dense_wgrad', x[r0:r0 + r], g[r0:r0 + r], ctx.wgrad.view(-1), r, K, Nn, 1)
        if ctx.bgrad is not None:
            ws = _ws('colsum', N.call('mmseg_colsum_workspace_floats', R, Nn), x.device)
            N.call('mmseg_colsum', g, ctx.bgrad, ws, R, Nn, 1.0, 1)
        _grad_done(ctx.wgrad, ctx.bgrad)
        return (dx,) + (None,) * 7


def dense(x, w, bias=None, act=None, alpha=0.0, wgrad=None, bgrad=None, anchor=None):
    if wgrad is None and bgrad is None:
        anchor = None
    return _Dense.apply(x, anchor, w, bias, act, alpha, wgrad, bgrad)


# ------------------------------------------------------------------------------------------------------
# FiLM (+LeakyReLU + residual add)
# ------------------------------------------------------------------------------------------------------
class _Film(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, alpha):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        B, H, W, C = x.shape
        assert gamma.shape == (B, C) and beta.shape == (B, C)
        y = _new(x.shape, x)
        N.call('mmseg_film_fwd', x, gamma, beta, _c(res) if res is not None else None, y, B, H * W, C, float(alpha))
        ctx.alpha, ctx.has_res = alpha, res is not None
        ctx.save_for_backward(x, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = _c(dy)
        dx = _new(x.shape, x)
        dg, db = _new((B, C), x), _new((B, C), x)
        ws = _ws('film', N.call('mmseg_film_bwd_workspace', B, C), x.device)
        N.call('mmseg_film_bwd', dy, x, gamma, beta, dx, dg, db, ws, B, H * W, C, float(ctx.alpha))
        return dx, dg, db, (dy if ctx.has_res else None), None


def film(x, gamma, beta, res=None, alpha=0.3):
    """leaky_relu(x * gamma + beta, alpha) [+ res]  (layers/film.py:26-36 + decoder.py:50-53)."""
    return _Film.apply(x, gamma, beta, res, alpha)


# ------------------------------------------------------------------------------------------------------
# thin-plate-spline warp, maximum, slice, sampling
# ------------------------------------------------------------------------------------------------------
class _TpsWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, theta, Mb):
        vol, theta = _c(vol), _c(theta)
        B, H, W, C = vol.shape
        assert theta.numel() == B * 50 and Mb.shape == (H * W, 25)
        out = _new(vol.shape, vol)
        loc = _new((B, H * W, 2), vol)
        N.call('mmseg_tps_warp_fwd', vol, theta, Mb, out, loc, B, H, W, C)
        ctx.save_for_backward(vol, loc, Mb)
        ctx.theta_shape = theta.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        vol, loc, Mb = ctx.saved_tensors
        B, H, W, C = vol.shape
        dout = _c(dout)
        dvol = dtheta = None
        acc = None
        if ctx.needs_input_grad[0]:
            dvol = _new(vol.shape, vol)
            acc = _ws('tps_acc', N.call('mmseg_tps_scatter_workspace_floats', B, H, W, C), vol.device)
        dloc = None
        ws = None
        if ctx.needs_input_grad[1]:
            dtheta = _new(ctx.theta_shape, vol)
            dloc = _ws('tps_dloc', B * H * W * 2, vol.device)
            ws = _ws('tps', N.call('mmseg_tps_workspace_floats', B), vol.device)
        N.call('mmseg_tps_warp_bwd', vol, loc, Mb, dout, dvol, dtheta, dloc, ws, acc, B, H, W, C)
        return dvol, dtheta, None


def tps_warp(vol, theta, Mb):
    return _TpsWarp.apply(vol, theta, Mb)


class _Maximum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = _new(a.shape, a)
        N.call('mmseg_maximum_fwd', a, b, y, a.numel())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        da = _new(a.shape, a) if ctx.needs_input_grad[0] else None
        db = _new(a.shape, a) if ctx.needs_input_grad[1] else None
        N.call('mmseg_maximum_bwd', a, b, _c(dy), da, db, a.numel())
        return da, db


def maximum(a, b):
    return _Maximum.apply(a, b)


class _SliceChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c0, cs):
        x = _c(x)
        C = x.shape[-1]
        y = _new(x.shape[:-1] + (cs,), x)
        N.call('mmseg_slice_fwd', x, y, x.numel() // C, C, c0, cs)
        ctx.meta = (x.shape, c0, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, c0, cs = ctx.meta
        dx = _new(shape, dy)
        N.call('mmseg_slice_bwd', _c(dy), dx, dx.numel() // shape[-1], shape[-1], c0, cs)
        return dx, None, None


def slice_channels(x, c0, cs):
    return _SliceChannels.apply(x, c0, cs)


class _SamplingKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, lv, eps):
        mu, lv, eps = _c(mu), _c(lv), _c(eps)
        B, Z = mu.shape
        z, kl = _new((B, Z), mu), _new((B, 1), mu)
        N.call('mmseg_sampling_kl_fwd', mu, lv, eps, z, kl, B, Z)
        ctx.save_for_backward(mu, lv, eps)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        mu, lv, eps = ctx.saved_tensors
        B, Z = mu.shape
        dmu, dlv = _new((B, Z), mu), _new((B, Z), mu)
        N.call('mmseg_sampling_kl_bwd', mu, lv, eps, _c(dz) if dz is not None else None,
               _c(dkl) if dkl is not None else None, dmu, dlv, B, Z)
        return dmu, dlv, None


def sampling_kl(z_mean, z_log_var, eps):
    """-> (z, kl[B,1])  (utils/sdnet_utils.py:9-21 with explicit eps, costs.py:186-189)."""
    return _SamplingKL.apply(z_mean, z_log_var, eps)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = _new(a.shape, a)
        N.call('mmseg_axpby', a, b, y, a.numel(), 1.0, 1.0)
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


# ------------------------------------------------------------------------------------------------------
# instance norm + SPADE modulation + LeakyReLU
# ------------------------------------------------------------------------------------------------------
class _InstNormSpade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, act_alpha):
        x = _c(x)
        B = x.shape[0]
        per = x.numel() // B
        gamma = _c(gamma) if gamma is not None else None
        beta = _c(beta) if beta is not None else None
        y = _new(x.shape, x)
        stat = _new((B, 2), x)
        ws = _ws('instnorm', N.call('mmseg_in_workspace_floats', B), x.device)
        N.call('mmseg_instnorm_spade_fwd', x, gamma, beta, y, stat, ws, B, per, IN_EPS, float(act_alpha))
        ctx.act_alpha = act_alpha
        ctx.save_for_backward(x, stat, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stat, gamma, beta = ctx.saved_tensors
        B = x.shape[0]
        per = x.numel() // B
        dx = _new(x.shape, x)
        dg = _new(x.shape, x) if gamma is not None else None
        db = _new(x.shape, x) if gamma is not None else None
        dxn = _ws('instnorm_dxn', x.numel(), x.device)
        ws = _ws('instnorm', N.call('mmseg_in_workspace_floats', B), x.device)
        N.call('mmseg_instnorm_spade_bwd', _c(dy), x, stat, gamma, beta, dx, dg, db, dxn, ws, B, per, IN_EPS,
               float(ctx.act_alpha))
        return dx, dg, db, None


# ------------------------------------------------------------------------------------------------------
# two convolutions of the same input as ONE (the gamma and beta convolutions of a SPADE unit)
# ------------------------------------------------------------------------------------------------------
_pair_cache = {}


def _pair_operands(wa, ba, wb, bb, wkey):
    """[KH, KW, Cin, Ca + Cb] kernel and [Ca + Cb] bias of the fused convolution: the two Keras kernels side by side.  Parameters of
    an nn.Model (`wkey`): a copy per weight version into buffers that live as long as the layer, so that the cached weight images
    keyed on them stay valid.  Anonymous weights (`wkey` None) are NEVER cached -- neither an address nor an object id identifies a
    tensor once the caching allocator has recycled it (round-3 review: a later weight of the same byte size got the previous
    layer's operands): they are concatenated on every call into scratch of the stream, like _wprep does."""
    KH, KW, Cin, Ca = wa.shape
    Cb = wb.shape[3]
    assert wb.shape == (KH, KW, Cin, Cb) and ba.numel() == Ca and bb.numel() == Cb
    nw, nb = KH * KW * Cin * (Ca + Cb), Ca + Cb
    if wkey is None:
        buf = _ws('pair_operands', nw + nb + 4, wa.device)
        wcat = buf[:nw].view(KH, KW, Cin, Ca + Cb)
        bcat = buf[(nw + 3) // 4 * 4:(nw + 3) // 4 * 4 + nb]
        N.call('mmseg_concat_cols', wa, wb, wcat, KH * KW * Cin, Ca, Cb)
        N.call('mmseg_concat_cols', ba, bb, bcat, 1, Ca, Cb)
        return wcat, bcat
    key = (wkey, wa.data_ptr(), wb.data_ptr(), _sid(wa.device))
    ent = _pair_cache.get(key)
    if ent is not None and (tuple(ent[1].shape) != (KH, KW, Cin, Ca + Cb) or ent[2].numel() != nb):
        ent = None           # never hand out operands of another geometry
    if ent is not None and ent[0] == _wver(wkey):
        return ent[1], ent[2]
    if ent is None:
        wcat = torch.empty((KH, KW, Cin, Ca + Cb), dtype=torch.float32, device=wa.device)
        bcat = torch.empty((Ca + Cb,), dtype=torch.float32, device=wa.device)
    else:
        wcat, bcat = ent[1], ent[2]
    N.call('mmseg_concat_cols', wa, wb, wcat, KH * KW * Cin, Ca, Cb)
    N.call('mmseg_concat_cols', ba, bb, bcat, 1, Ca, Cb)
    _pair_cache[key] = (_wver(wkey), wcat, bcat)
    return wcat, bcat


class _Conv2dPair(torch.autograd.Function):
    """y[..., :Ca] = conv(x, wa) + ba, y[..., Ca:] = conv(x, wb) + bb (3x3 'same', stride 1) as ONE convolution with Ca + Cb output
    channels: x is read once, the 32-wide tiles of two narrow convolutions become one 64-wide tile, and in the backward pass one
    weight-gradient and one data-gradient launch replace two of each (plus the addition of the two data gradients).  Per output
    channel the arithmetic is that of the separate convolutions."""

    @staticmethod
    def forward(ctx, x, anchor, wa, ba, wb, bb, grads, wkey, out_dtype=torch.float32):
        x = _c(x)
        B, H, W, Cin = x.shape
        KH, KW, Cin2, Ca = wa.shape
        Cb = wb.shape[3]
        assert Cin == Cin2 and wb.shape[:3] == wa.shape[:3] and KH % 2 == 1 and KW % 2 == 1
        Cout = Ca + Cb
        wcat, bcat = _pair_operands(wa, ba, wb, bb, wkey)
        pkey = (('pair',) + tuple(wkey[:1]), wkey[1]) if isinstance(wkey, tuple) else None
        wt = _wprep(wcat, KH, KW, Cin, Cout, 0, pkey) if N.call('mmseg_conv2d_fast_path', Cin, 0, Cout, 0) else None
        # 16-bit outputs need the fast path in BOTH directions: the backward pass reads the 16-bit dy and writes a 16-bit dx through
        # the fast path of the Cout -> Cin convolution (advisor, round 3: 2f % 32 != 0 sent a 16-bit dy to the generic kernel)
        bwd_fast = bool(N.call('mmseg_conv2d_fast_path', Cout, 0, Cin, 0))
        y = _new((B, H, W, Cout), x, out_dtype if (wt is not None and bwd_fast) else torch.float32)
        _conv_fwd_raw(x, None, wcat, wt, bcat, y, None, B, H, W, Cin, 0, H, W, Cout, KH, KW, 1, KH // 2, KW // 2, 0, 0, 0, 0.0, 0)
        ctx.geom = (B, H, W, Cin, Ca, Cb, KH, KW)
        ctx.grads, ctx.pkey = grads, pkey
        ctx.operands = (wa, ba, wb, bb, wkey)     # plain (non-leaf) weight views: not tracked by autograd
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, Cin, Ca, Cb, KH, KW = ctx.geom
        Cout, M, K = Ca + Cb, B * H * W, KH * KW * Cin
        dy = _c(dy)
        wga, bga, wgb, bgb = ctx.grads
        if bga is not None:
            tmp = _ws('pair_db', Cout, dy.device)[:Cout]
            ws = _ws('colsum', N.call('mmseg_colsum_workspace_floats', M, Cout), dy.device)
            if _h(dy) and Cout % 64 == 0:
                N.call('mmseg_colsum_t', dy, tmp, ws, M, Cout, 0, _h(dy))
            else:
                N.call('mmseg_colsum', dy if _h(dy) == 0 else dy.float(), tmp, ws, M, Cout, 1.0, 0)
            N.call('mmseg_split_cols_acc', tmp, bga, bgb, 1, Ca, Cb)
        if wga is not None:
            tmp = _ws('pair_dw', K * Cout, dy.device)[:K * Cout]
            _wgrad_launch(x, None, dy, tmp, B, H, W, Cin, 0, H, W, Cout, KH, KW, 1, KH // 2, KW // 2, 0, 0)
            N.call('mmseg_split_cols_acc', tmp, wga.view(-1), wgb.view(-1), K, Ca, Cb)
        _grad_done(wga, bga, wgb, bgb)
        dx = None
        if ctx.needs_input_grad[0]:
            # (a model's operands come out of the cache; anonymous ones are concatenated again: their scratch may have been reused)
            wcat = _pair_operands(*ctx.operands)[0]
            fastb = bool(N.call('mmseg_conv2d_fast_path', Cout, 0, Cin, 0))
            if not fastb and _h(dy):
                dy = dy.float()                             # (unreachable with the forward's choice of out_dtype; kept as a guard)
            dx = _new((B, H, W, Cin), dy, x.dtype if (fastb or _h(x) == 0) else torch.float32)      # a gradient is stored like its tensor
            if fastb:
                wf, wt = None, _wprep(wcat, KH, KW, Cin, Cout, 1, ctx.pkey)
            else:
                wf, wt = _ws('wflip', wcat.numel(), dy.device)[:wcat.numel()], None
                N.call('mmseg_conv2d_wflip', wcat, wf, KH, KW, Cin, Cout)
            _conv_fwd_raw(dy, None, wf, wt, None, dx, None, B, H, W, Cout, 0, H, W, Cin, KH, KW, 1, KH - 1 - KH // 2, KW - 1 - KW // 2,
                          0, 0, 0, 0.0, 0)
        return (dx,) + (None,) * 8


def conv2d_pair(x, wa, ba, wb, bb, grads=(None, None, None, None), anchor=None, wkey=None, out_dtype=torch.float32):
    if all(g is None for g in grads):
        anchor = None
    return _Conv2dPair.apply(x, anchor, wa, ba, wb, bb, tuple(grads), wkey, out_dtype)


class _InstNormSpadeGB(torch.autograd.Function):
    """instnorm_spade with gamma and beta as the halves of one tensor gb [B, H, W, 2C] (the output of conv2d_pair)"""

    @staticmethod
    def forward(ctx, x, gb, act_alpha, out_dtype=torch.float32):
        x, gb = _c(x), _c(gb)
        B, C = x.shape[0], x.shape[-1]
        assert gb.shape == x.shape[:-1] + (2 * C,)
        per = x.numel() // B
        y = _new(x.shape, x, out_dtype)
        stat = _new((B, 2), x)
        ws = _ws('instnorm', N.call('mmseg_in_workspace_floats', B), x.device)
        N.call('mmseg_instnorm_spade_fwd_gb_t', x, gb, y, stat, ws, B, per, C, IN_EPS, float(act_alpha), _h(gb), _h(y))
        ctx.act_alpha = act_alpha
        ctx.save_for_backward(x, stat, gb)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stat, gb = ctx.saved_tensors
        B, C = x.shape[0], x.shape[-1]
        per = x.numel() // B
        dx, dgb = _new(x.shape, x), _new(gb.shape, gb, gb.dtype)          # a gradient is stored like its tensor
        dxn = _ws('instnorm_dxn', x.numel(), x.device)
        ws = _ws('instnorm', N.call('mmseg_in_workspace_floats', B), x.device)
        dy = _c(dy)
        N.call('mmseg_instnorm_spade_bwd_gb_t', dy, x, stat, gb, dx, dgb, dxn, ws, B, per, C, IN_EPS, float(ctx.act_alpha), _h(gb), _h(dy))
        return dx, dgb, None, None


def instnorm_spade_gb(x, gb, act_alpha=-1.0, out_dtype=torch.float32):
    """act( IN(x) * (1 + gb[..., :C]) + gb[..., C:] ); `out_dtype`: storage type of the result (and of its gradient)"""
    return _InstNormSpadeGB.apply(x, gb, act_alpha, out_dtype)


class _ExpandScalar(torch.autograd.Function):
    """A 1-element parameter (the scalar gamma / beta of keras_contrib InstanceNormalization(axis=None)) as a [B, C] table
    for the FiLM kernel; the backward pass sums the table's gradient into the parameter's gradient-arena slot."""

    @staticmethod
    def forward(ctx, anchor, p, B, C, grad):
        ctx.grad = grad
        return p.reshape(1, 1).expand(B, C).contiguous()

    @staticmethod
    def backward(ctx, dg):
        if ctx.grad is not None:
            dg = _c(dg)
            ws = _ws('colsum', N.call('mmseg_colsum_workspace_floats', dg.numel(), 1), dg.device)
            N.call('mmseg_colsum', dg.reshape(-1), ctx.grad, ws, dg.numel(), 1, 1.0, 1)
        _grad_done(ctx.grad)
        return (None,) * 5


def expand_scalar(p, B, C, grad=None, anchor=None):
    return _ExpandScalar.apply(anchor if grad is not None else None, p, B, C, grad)


def instnorm_spade(x, gamma=None, beta=None, act_alpha=-1.0):
    """act( IN(x) * (1 + gamma) + beta ); act_alpha < 0: no activation (layers/spade.py:7-33,51-54)."""
    return _InstNormSpade.apply(x, gamma, beta, act_alpha)


# ------------------------------------------------------------------------------------------------------
# losses: value (device scalar) + gradient w.r.t. the prediction, no autograd node
# ------------------------------------------------------------------------------------------------------
def seg_loss(pred, target, num_masks, lambda_bce, scale, class_sum_hook=None, n_pix_global=None, want_grad=True):
    """make_combined_dice_bce (lambda_bce = 0.01) or make_dice_loss_fnc (lambda_bce = 0) of costs.py.
    `class_sum_hook(t)` may all-reduce the batch-global class sums in place (data-parallel training).
    -> (loss[1], dpred or None); dpred already multiplied by `scale` (loss weight)."""
    pred, target = _c(pred), _c(target)
    B, H, W, C = pred.shape
    dev = pred.device
    stats = _new((N.call('mmseg_segloss_stats_floats', B),), pred)
    ws = _ws('segloss', N.call('mmseg_segloss_workspace_floats', B), dev)
    N.call('mmseg_segloss_stats', pred, target, stats, ws, B, H * W, C, num_masks)
    if class_sum_hook is not None and lambda_bce != 0.0:
        class_sum_hook(stats[N.call('mmseg_segloss_class_offset', B):])
    loss = _new((1,), pred)
    coef = _new((N.call('mmseg_segloss_coef_floats', B, C),), pred)
    npl = float(B * H * W)
    npg = npl if n_pix_global is None else float(n_pix_global)
    # loss value over the global pixel count; gradient seeds over the LOCAL one (the DP all-reduce averages over ranks)
    N.call('mmseg_segloss_finalize', stats, loss, coef, B, C, npg, npl, float(lambda_bce))
    dpred = None
    if want_grad:
        dpred = _new(pred.shape, pred)
        N.call('mmseg_segloss_grad', pred, target, coef, dpred, B, H * W, C, num_masks, float(scale),
               int(lambda_bce != 0.0))
    return loss, dpred


# ------------------------------------------------------------------------------------------------------
# in-graph per-sample loss terms of the automated-pairing trainers (csrc/pairloss.hip)
# ------------------------------------------------------------------------------------------------------
class _OverlapDice(torch.autograd.Function):
    """balancer.dice of one reference anatomy against J others -> [B, J] (model_components/balancer.py:21-22,33-38)"""

    @staticmethod
    def forward(ctx, ref, *others):
        ref = _c(ref)
        others = [_c(o) for o in others]
        B, J = ref.shape[0], len(others)
        per = ref.numel() // B
        out = _new((B, J), ref)
        stats = _new((J, B, 3), ref)
        ws = _ws('pairloss', N.call('mmseg_pairloss_workspace_floats', B), ref.device)
        for j, o in enumerate(others):
            N.call('mmseg_pair_dice_fwd', ref, o, stats[j], _col(out, j),
                   J, ws, B, per)
        ctx.save_for_backward(ref, stats, *others)
        return out

    @staticmethod
    def backward(ctx, g):
        ref, stats = ctx.saved_tensors[:2]
        others = ctx.saved_tensors[2:]
        g = _c(g)
        B, J = g.shape
        per = ref.numel() // B
        dref = _new(ref.shape, ref)
        douts = []
        for j, o in enumerate(others):
            do = _new(o.shape, o) if ctx.needs_input_grad[1 + j] else None
            N.call('mmseg_pair_dice_bwd', ref, o, stats[j], _col(g, j), J, dref, do, int(j > 0), B, per)
            douts.append(do)
        return (dref if ctx.needs_input_grad[0] else None,) + tuple(douts)


def _col(t, j):
    """view of t[:, j:] flattened so that element [b, j] sits at offset b * row_stride (used with an explicit ld)"""
    flat = t.reshape(-1)
    return flat[j:]


def overlap_dice(ref, others):
    return _OverlapDice.apply(ref, *others)


class _RowMAE(torch.autograd.Function):
    """costs.mae_single_input: mean |x - y| over all but the batch axis -> [B]; gradient to y only (x is data)"""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        B = y.shape[0]
        per = y.numel() // B
        out = _new((B,), y)
        ws = _ws('pairloss', N.call('mmseg_pairloss_workspace_floats', B), y.device)
        N.call('mmseg_row_mae_fwd', x, y, out, ws, B, per)
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B = y.shape[0]
        dy = _new(y.shape, y)
        N.call('mmseg_row_mae_bwd', x, y, _c(g), dy, B, y.numel() // B)
        return None, dy


def row_mae(x, y):
    return _RowMAE.apply(x, y)


class _SegLossPerSample(torch.autograd.Function):
    """costs.make_combined_dice_bce_perbatch(num_masks)(y_true, y_pred) -> [B] (costs.py:138-143), including the
    swapped-argument cross-entropy (class weights from the prediction over the whole batch, log-softmax of the labels)."""

    @staticmethod
    def forward(ctx, target, pred, num_masks, lambda_bce, class_sum_hook):
        target, pred = _c(target), _c(pred)
        B, H, W, C = pred.shape
        stats = _new((N.call('mmseg_segpb_stats_floats', B),), pred)
        ws = _ws('pairloss', N.call('mmseg_pairloss_workspace_floats', B), pred.device)
        N.call('mmseg_segpb_stats', pred, target, stats, ws, B, H * W, C, num_masks)
        if class_sum_hook is not None:
            class_sum_hook(stats[N.call('mmseg_segpb_class_offset', B):])
        loss = _new((B,), pred)
        N.call('mmseg_segpb_loss', stats, loss, B, H * W, C, float(lambda_bce))
        ctx.save_for_backward(target, stats)
        ctx.meta = (B, H * W, C, num_masks, float(lambda_bce), class_sum_hook)
        return loss

    @staticmethod
    def backward(ctx, g):
        target, stats = ctx.saved_tensors
        B, HW, C, nm, lam, hook = ctx.meta
        g = _c(g)
        A = _new((8,), g)
        N.call('mmseg_segpb_classgrad', stats, g, A, B)
        if hook is not None:
            hook(A)
        dpred = _new(target.shape, target)
        N.call('mmseg_segpb_grad', target, stats, g, A, dpred, B, HW, C, nm, lam)
        return None, dpred, None, None, None


def seg_loss_per_sample(target, pred, num_masks, lambda_bce=0.01, class_sum_hook=None):
    return _SegLossPerSample.apply(target, pred, num_masks, lambda_bce, class_sum_hook)


class _RowDot(torch.autograd.Function):
    """sum_j w[:, j] * l_j -> [B, 1]: keras Multiply + Add over the candidate pairs (models/dafnet.py:293-312)"""

    @staticmethod
    def forward(ctx, w, *ls):
        w = _c(w)
        B, J = w.shape
        l = torch.stack([x.reshape(B) for x in ls], dim=1).contiguous()      # [B, J] gather of J*B scalars
        out = _new((B, 1), w)
        N.call('mmseg_rowdot_fwd', w, l, out, B, J)
        ctx.save_for_backward(w, l)
        ctx.shapes = [tuple(x.shape) for x in ls]
        return out

    @staticmethod
    def backward(ctx, g):
        w, l = ctx.saved_tensors
        B, J = w.shape
        dw, dl = _new(w.shape, w), _new(l.shape, l)
        N.call('mmseg_rowdot_bwd', w, l, _c(g), dw, dl, B, J)
        return (dw,) + tuple(dl[:, j].reshape(shp) for j, shp in enumerate(ctx.shapes))


def row_dot(w, ls):
    return _RowDot.apply(w, *ls)


_DIFF_MODE = {'mae': 0, 'mse': 1, 'mean': 2}


def diff_loss(pred, target, mode, scale, want_grad=True):
    """keras 'mae' / 'mse' (target tensor or python float) or mean(pred) (costs.ypred).
    -> (loss[1], dpred or None), dpred = scale * dloss/dpred."""
    pred = _c(pred)
    n = pred.numel()
    t, tc = (None, float(target)) if not isinstance(target, torch.Tensor) else (_c(target), 0.0)
    if t is not None:
        assert t.numel() == n
    loss = _new((1,), pred)
    ws = _ws('diffloss', N.call('mmseg_diffloss_workspace_floats'), pred.device)
    N.call('mmseg_diffloss', pred, t, tc, n, _DIFF_MODE[mode], loss, ws)
    dpred = None
    if want_grad:
        dpred = _new(pred.shape, pred)
        g = {0: 1.0 / n, 1: 1.0 / n, 2: 1.0 / n}[_DIFF_MODE[mode]] * scale
        N.call('mmseg_diffloss_grad', pred, t, tc, n, _DIFF_MODE[mode], float(g), dpred)
    return loss, dpred


def spectral_reg(w, u0, alpha=10.0):
    """layers/spectralnorm.py:216-239 -> (loss[1], sgn[1]); gradient via spectral_reg_grad."""
    K = w.numel() // w.shape[-1]
    Nn = w.shape[-1]
    assert u0.numel() == K
    loss, sgn = _new((1,), w), _new((1,), w)
    ws = _ws('spectral', N.call('mmseg_spectral_workspace_floats', K, Nn), w.device)
    N.call('mmseg_spectral_fwd', w, u0, loss, sgn, ws, K, Nn, float(alpha))
    return loss, sgn


def spectral_reg_multi(ws_, u0s, alpha=10.0):
    """The Spectral penalties of up to 4 kernels in one batch of launches -> (loss[n], sgn[n])."""
    n = len(ws_)
    assert 1 <= n <= 4 and len(u0s) == n
    dims = [(w.numel() // w.shape[-1], w.shape[-1]) for w in ws_]
    loss, sgn = _new((n,), ws_[0]), _new((n,), ws_[0])
    need = sum(N.call('mmseg_spectral_workspace_floats', K, Nn) for K, Nn in dims)
    ws = _ws('spectral', need, ws_[0].device)
    pad = lambda lst, fill: list(lst) + [fill] * (4 - n)
    flat = [v for d in pad(dims, dims[0]) for v in d]
    N.call('mmseg_spectral_fwd4', *pad(ws_, ws_[0]), *pad(u0s, u0s[0]), loss, sgn, ws, n, *flat, float(alpha))
    return loss, sgn


def spectral_reg_grad_accumulate(ws_, sgn, grads, scale=1.0):
    """grads[i] += scale * d penalty_i / d W_i  (one launch)"""
    n = len(ws_)
    pad = lambda lst, fill: list(lst) + [fill] * (4 - n)
    N.call('mmseg_spectral_grad4', *pad(ws_, ws_[0]), sgn, *pad(grads, grads[0]), n, *pad([w.numel() for w in ws_], 0), float(scale))


def spectral_reg_grad(w, sgn, scale=1.0):
    dw = _new(w.shape, w)
    N.call('mmseg_spectral_grad', w, sgn, float(scale), w.numel(), dw)
    return dw


def adam_step(p, g, m, v, lr_t, beta_1=0.9, beta_2=0.999, eps=1e-7, owner=None):
    """Keras 2.1.6 Adam over flat arenas (p, g, m, v: 1-D views of equal length).  `owner`: id of the nn.Model that owns the arena
    (its cached weight images become stale; None: all of them)."""
    assert p.numel() == g.numel() == m.numel() == v.numel()
    if isinstance(lr_t, torch.Tensor):       # a device scalar: the launch stays valid when replayed from a captured graph
        N.call('mmseg_adam_p', p, g, m, v, p.numel(), lr_t, float(beta_1), float(beta_2), float(eps))
    else:
        N.call('mmseg_adam', p, g, m, v, p.numel(), float(lr_t), float(beta_1), float(beta_2), float(eps))
    bump_weight_version(owner)


def fill_(t, value):
    N.call('mmseg_fill', t, t.numel(), float(value))
    return t


def axpby(a, b, sa=1.0, sb=1.0, out=None):
    out = _new(a.shape, a) if out is None else out
    N.call('mmseg_axpby', _c(a), _c(b), out, a.numel(), float(sa), float(sb))
    return out


def affine_gather(data, rows, mat, order=1):
    """out[b] = affine resample (order 0/1, edge-clamped) of data[rows[b]] with the 2x3 matrix mat[b] in (row, col)
    coordinates -- batch assembly + augmentation in one launch (csrc/augment.hip).  No gradient (input pipeline)."""
    B = mat.shape[0]
    out = _new((B,) + tuple(data.shape[1:]), data)
    N.call('mmseg_affine_gather', data, rows, _c(mat), out, B, data.shape[1], data.shape[2], data.shape[3], int(order))
    return out


def _sum_n(gs, like):
    """sum of 1..n same-shaped tensors in as few launches as possible (8 operands per launch, left to right)"""
    gs = [_c(g) for g in gs]
    while len(gs) > 1:
        grp, rest = gs[:8], gs[8:]
        out = _new(like.shape, like, grp[0].dtype)
        assert all(g.dtype == grp[0].dtype and g.shape == grp[0].shape for g in grp)
        N.call('mmseg_sum_n_t', *(grp + [None] * (8 - len(grp))), len(grp), out, out.numel(), _h(out))
        gs = [out] + rest
    return gs[0]


class _Share(torch.autograd.Function):
    """n aliases of x for n consumers.  The autograd engine adds the gradients of a tensor with several consumers pairwise with torch
    kernels (46 additions per DAFNet generator step; 6 of them chained for the anatomy factor alone); consumers that take their
    own alias meet again HERE, where one launch adds all of them (mmseg_sum_n_t)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        return _sum_n(gs, gs[0]), None


class Shared(object):
    """`s = Shared(x, n)`; every consumer calls `s.use()` (the n-th use hands out the last alias; more uses raise).  Without a tape
    (inference) it hands out x itself."""

    def __init__(self, x, n):
        self.x, self.n, self.i = x, n, 0
        self.parts = _Share.apply(x, n) if (n > 1 and torch.is_grad_enabled() and x.requires_grad) else None

    def use(self):
        if self.parts is None:
            return self.x
        if self.i >= self.n:
            raise RuntimeError('Shared tensor declared for %d consumers is used a %d-th time' % (self.n, self.i + 1))
        self.i += 1
        return self.parts[self.i - 1]


def share(x, n):
    """-> list of n aliases of x, one per consumer (see _Share)"""
    sh = Shared(x, n)
    return [sh.use() for _ in range(n)]


def _cat_words(parts, out):
    """out (contiguous, n * len(part) rows) <- parts stacked on the leading axis; a None part is written as zeros"""
    n = len(parts)
    B = out.shape[0] // n
    words = (out.numel() // n) * out.element_size()
    assert words % 4 == 0, 'batch concatenation moves 4-byte words'
    words //= 4
    for k0 in range(0, n, 8):
        grp = parts[k0:k0 + 8]
        N.call('mmseg_cat_words', *(list(grp) + [None] * (8 - len(grp))), len(grp), out[k0 * B:(k0 + len(grp)) * B], words)


class _CatBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        ts = [_c(t) for t in ts]
        t0 = ts[0]
        assert all(t.shape == t0.shape and t.dtype == t0.dtype for t in ts), 'cat_batch: equally shaped parts'
        out = _new((len(ts) * t0.shape[0],) + tuple(t0.shape[1:]), t0, t0.dtype)
        _cat_words(ts, out)
        ctx.n, ctx.B = len(ts), t0.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return tuple(g[i * ctx.B:(i + 1) * ctx.B] for i in range(ctx.n))      # contiguous row blocks: views, nothing is copied


class _SplitBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        x = _c(x)
        assert x.shape[0] % n == 0
        B = x.shape[0] // n
        ctx.meta = (tuple(x.shape), x.dtype, n)
        return tuple(x[i * B:(i + 1) * B] for i in range(n))

    @staticmethod
    def backward(ctx, *gs):
        shape, dtype, n = ctx.meta
        like = next((g for g in gs if g is not None), None)
        if like is None:
            return None, None
        out = torch.empty(shape, dtype=dtype, device=like.device)
        _cat_words([_c(g) if g is not None else None for g in gs], out)
        return out, None


def cat_batch(tensors):
    """Stack independent calls of a per-sample component on the batch axis (pure data movement in one launch, mmseg_cat_words; the
    backward pass hands every producer its row block of the gradient as a view).  Components without batch statistics --
    discriminators, FiLM / SPADE decoders, the anatomy fuser, inference-mode BatchNorm -- give identical per-sample results, with
    1/n of the launches and GEMMs large enough to fill the chip."""
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    return _CatBatch.apply(*tensors)


def split_batch(t, n):
    """inverse of cat_batch for n equal parts (views; their gradients are gathered into one tensor by one launch)"""
    if n == 1:
        return [t]
    return list(_SplitBatch.apply(t, n))


def gather_rows(src, idx):
    """src[idx] along the leading axis (idx: int64 device tensor): sampling a fake pool (utils/data_utils.py sample of the reference)"""
    src = _c(src)
    rows = idx.numel()
    out = _new((rows,) + tuple(src.shape[1:]), src, src.dtype)
    row_bytes = (src.numel() // max(src.shape[0], 1)) * src.element_size()
    assert row_bytes % 4 == 0
    N.call('mmseg_gather_rows', src, idx, out, rows, row_bytes // 4, src.shape[0])
    return out


def add_residual(m):
    """[..., C] masks -> [..., C + 1]: background channel = 1 unless some mask equals 1 exactly (base_executor.py:83-87)"""
    m = _c(m)
    C = m.shape[-1]
    out = _new(tuple(m.shape[:-1]) + (C + 1,), m)
    N.call('mmseg_add_residual', m, out, m.numel() // C, C)
    return out
