"""TEST-ONLY stand-in for libmmseg_hip.so on machines without a GPU.

`install()` replaces multimodal_segmentation_amd._native.call with a dispatcher that executes each C-ABI entry
point with plain torch-CPU ops on the SAME argument list (tensors instead of device pointers), writing into the
caller's output tensors.  It exists so that the host logic above the C ABI (autograd nodes, model wiring, executors,
the data-parallel path under gloo) can be exercised by `pytest -m "not gpu"`.  It is never imported by the product
package, is not a fallback, and is not what any parity claim rests on: GPU parity tests call the real library.
"""
import math

import torch
import torch.nn.functional as F

from oracle import ops as O


def _act(v, act, alpha):
    if act == 1:
        return torch.relu(v)
    if act == 2:
        return torch.where(v >= 0, v, v * alpha)
    if act == 3:
        return torch.tanh(v)
    return v


def _act_grad(y, act, alpha):
    if act == 1:
        return (y > 0).to(y.dtype)
    if act == 2:
        return torch.where(y >= 0, torch.ones_like(y), torch.full_like(y, alpha))
    if act == 3:
        return 1 - y * y
    return torch.ones_like(y)


def _logical_input(x1, x2, B, H, W, C1, C2, ups):
    a = x1.reshape(B, H >> ups, W >> ups, C1)
    if ups:
        a = a.repeat_interleave(2, 1).repeat_interleave(2, 2)
    if C2:
        a = torch.cat([a, x2.reshape(B, H, W, C2)], -1)
    return a


def conv2d_fast_path(C1, C2, Cout, transposed):
    return 0


_precision = [0]


def set_conv_precision(mode):
    old = _precision[0]
    _precision[0] = mode if mode in (1, 2) else 0
    return old


def get_conv_precision():
    return _precision[0]


def conv16_mode(mode):
    return 1


def conv8h_fwd_t(x, w, bias, y, B, H, W, Cout, act, alpha, hx, hy):
    """Conv2D(Cout, 3, 'same') of an 8-channel tensor (16-bit modes of the HIP library; here plain fp32 arithmetic)"""
    wk = w.reshape(3, 3, 8, Cout).permute(3, 2, 0, 1).float()
    v = torch.nn.functional.conv2d(x.reshape(B, H, W, 8).float().permute(0, 3, 1, 2), wk, None if bias is None else bias.float(), padding=1)
    y.copy_(_act(v, act, alpha).permute(0, 2, 3, 1).reshape(y.shape).to(y.dtype)); return 0


def conv2d_parity_taps(K, stride, p):
    return (K - p + stride - 1) // stride if p < K else 0


def conv2d_wprep(w, out, KH, KW, Cin, Cout, mode):
    wk = w.reshape(KH * KW, Cin, Cout)
    r = wk.permute(2, 0, 1) if mode == 0 else torch.flip(wk, (0,)).permute(1, 0, 2)
    out.copy_(r.reshape(-1)); return 0


def conv2d_fwd(x1, x2, w, wt, bias, y, y2, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, transposed, act,
               alpha, nsplit1):
    xin = _logical_input(x1, x2, B, H, W, C1, C2, ups).permute(0, 3, 1, 2)
    if w is None:      # only the fast layout was supplied: [Cout][KH*KW][Cin]
        w = wt.reshape(Cout, KH * KW, C1 + C2).permute(1, 2, 0)
    wk = w.reshape(KH, KW, C1 + C2, Cout).permute(3, 2, 0, 1)
    if transposed:
        # zero-dilate the input by `stride`, then a stride-1 correlation with padding (ph, pw), cropped to Ho x Wo
        z = torch.zeros(B, C1 + C2, (H - 1) * stride + 1, (W - 1) * stride + 1, dtype=xin.dtype)
        z[:, :, ::stride, ::stride] = xin
        need_h, need_w = Ho + KH - 1, Wo + KW - 1
        z = F.pad(z, (pw, max(0, need_w - pw - z.shape[3]), ph, max(0, need_h - ph - z.shape[2])))
        out = F.conv2d(z, wk)[:, :, :Ho, :Wo]
    else:
        out = F.conv2d(xin, wk, None, stride=stride, padding=(ph, pw))
    out = out.permute(0, 2, 3, 1)
    if bias is not None:
        out = out + bias
    out = _act(out, act, alpha)
    if y2 is None:
        y.copy_(out.reshape(y.shape))
    else:
        y.copy_(out[..., :nsplit1].reshape(y.shape))
        y2.copy_(out[..., nsplit1:].reshape(y2.shape))
    return 0


def conv2d_fwd_scaled(x1, x2, w, wt, bias, oscale, y, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, act, alpha):
    tmp = torch.empty_like(y)
    conv2d_fwd(x1, x2, w, wt, None, tmp, None, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, 0, 0, 0.0, 0)
    v = tmp.reshape(-1, Cout) * oscale.reshape(1, Cout) + (bias.reshape(1, Cout) if bias is not None else 0.0)
    y.copy_(_act(v, act, alpha).reshape(y.shape))
    return 0


def bn_infer_fold(gamma, beta, mm, mv, conv_bias, scale, shift, C, eps):
    sc = gamma / torch.sqrt(mv + eps)
    scale.copy_(sc)
    shift.copy_(beta - mm * sc + (conv_bias * sc if conv_bias is not None else 0.0))
    return 0


def bn_stats_local(x, stat2, ws, M, C):
    xr = x.reshape(M, C)
    stat2.copy_(torch.stack([xr.mean(0), xr.var(0, unbiased=False)]).reshape(stat2.shape)); return 0


def bn_stats_combine(gathered, R, gamma, beta, mean, invstd, scale, shift, mov_mean, mov_var, M_total, C, eps, momentum):
    g = gathered.reshape(R, 2, C)
    mu = g[:, 0].mean(0)
    var = (g[:, 1] + (g[:, 0] - mu) ** 2).mean(0)
    is_ = torch.rsqrt(var + eps)
    mean.copy_(mu); invstd.copy_(is_); scale.copy_(gamma * is_); shift.copy_(beta - mu * gamma * is_)
    if mov_mean is not None:
        unb = var * (M_total / (M_total - 1.0)) if M_total > 1 else var
        mov_mean.sub_((mov_mean - mu) * (1.0 - momentum)); mov_var.sub_((mov_var - unb) * (1.0 - momentum))
    return 0


def bn_bwd_sums(dy, y, x, mean, invstd, sums, ws, M, C, relu):
    g = dy.reshape(M, C)
    if relu:
        g = g * (y.reshape(M, C) > 0)
    xh = (x.reshape(M, C) - mean) * invstd
    sums.copy_(torch.stack([g.sum(0), (g * xh).sum(0)]).reshape(sums.shape)); return 0


def bn_bwd_finish(local, glob, gamma, mean, invstd, dgamma, dbeta, coef, C, M_total, accumulate):
    l, g = local.reshape(2, C), glob.reshape(2, C)
    if dgamma is not None:
        if accumulate:
            dbeta.add_(l[0]); dgamma.add_(l[1])
        else:
            dbeta.copy_(l[0]); dgamma.copy_(l[1])
    A = gamma * invstd
    Bc = -gamma * invstd * invstd * g[1] / M_total
    Cc = -A * g[0] / M_total - Bc * mean
    coef[:3 * C].copy_(torch.cat([A, Bc, Cc])); return 0


def bn_bwd_apply(dy, y, x, coef, dx, M, C, relu):
    g = dy.reshape(M, C)
    if relu:
        g = g * (y.reshape(M, C) > 0)
    A, Bc, Cc = coef[:C], coef[C:2 * C], coef[2 * C:3 * C]
    dx.copy_((A * g + Bc * x.reshape(M, C) + Cc).reshape(dx.shape)); return 0


def conv2d_dgrad_tapsum(T, dx, B, H, W, Ho, Wo, Cin, KH, KW, ph, pw):
    t = T.reshape(B, Ho, Wo, KH * KW, Cin)
    out = torch.zeros(B, H, W, Cin, dtype=T.dtype)
    for kh in range(KH):
        for kw in range(KW):
            # dx[h, w] += T[h + ph - kh, w + pw - kw, tap]
            h0, h1 = max(0, kh - ph), min(H, Ho + kh - ph)
            w0, w1 = max(0, kw - pw), min(W, Wo + kw - pw)
            if h1 > h0 and w1 > w0:
                out[:, h0:h1, w0:w1] += t[:, h0 + ph - kh:h1 + ph - kh, w0 + pw - kw:w1 + pw - kw, kh * KW + kw]
    dx.copy_(out.reshape(dx.shape)); return 0


def conv2d_wgrad_workspace(B, Ho, Wo, Cin, Cout, KH, KW):
    return 0


def conv2d_wgrad(x1, x2, dy, dw, ws, ws_floats, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, ph, pw, ups, accumulate):
    xin = _logical_input(x1, x2, B, H, W, C1, C2, ups).permute(0, 3, 1, 2)
    g = dy.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(xin, (Cout, C1 + C2, KH, KW), g, stride=stride, padding=(ph, pw))
    gw = gw.permute(2, 3, 1, 0).reshape(dw.shape)
    if accumulate:
        dw.add_(gw)
    else:
        dw.copy_(gw)
    return 0


def conv2d_wflip(w, wt, KH, KW, Cin, Cout):
    wk = w.reshape(KH, KW, Cin, Cout)
    wt.copy_(torch.flip(wk, (0, 1)).permute(0, 1, 3, 2).reshape(wt.shape))
    return 0


def act_fwd(x, y, n, act, alpha):
    y.copy_(_act(x, act, alpha)); return 0


def act_bwd(dy, y, dx, n, act, alpha):
    dx.copy_(dy * _act_grad(y, act, alpha)); return 0


def act_bwd_bias_t(dy, y, dx, bias_grad, ws, M, C, act, alpha, accumulate, h):
    d = dy.float() * _act_grad(y.float(), act, alpha)
    dx.copy_(d.to(dx.dtype))
    if bias_grad is not None:
        s = dx.float().reshape(M, C).sum(0)
        bias_grad.copy_(bias_grad + s if accumulate else s)
    return 0


def axpby(a, b, out, n, sa, sb):
    out.copy_(a * sa + b * sb); return 0


def fill(x, n, v):
    x.fill_(v); return 0


def colsum_workspace_floats(M, C):
    return 1


def colsum_blocks(M):
    return 1


def colsum(x, out, ws, M, C, scale, accumulate):
    s = x.reshape(M, C).sum(0) * scale
    out.copy_(out + s if accumulate else s); return 0


def maxpool2_fwd(x, y, B, H, W, C):
    y.copy_(O.maxpool2(x.reshape(B, H, W, C))); return 0


def maxpool2_bwd(x, y, dy, dx, B, H, W, C):
    with torch.enable_grad():
        xr = x.reshape(B, H, W, C).detach().clone().requires_grad_(True)
        out = O.maxpool2(xr)
        (g,) = torch.autograd.grad(out, xr, dy.reshape(out.shape))
    dx.copy_(g); return 0


def maxpool2_bwd_add_t(x, y, dy, add, dx, B, H, W, C, h):
    maxpool2_bwd(x, y, dy, dx, B, H, W, C)
    if add is not None:
        dx.add_(add.reshape(dx.shape))
    return 0


def sum_n_t(p0, p1, p2, p3, p4, p5, p6, p7, n, out, numel, h):
    ps = [p0, p1, p2, p3, p4, p5, p6, p7][:n]
    acc = ps[0].reshape(-1).float().clone()
    for q in ps[1:]:
        acc += q.reshape(-1).float()
    out.copy_(acc.reshape(out.shape).to(out.dtype)); return 0


def cat_words(p0, p1, p2, p3, p4, p5, p6, p7, n, out, words):
    ps = [p0, p1, p2, p3, p4, p5, p6, p7][:n]
    B = out.shape[0] // n
    for k, q in enumerate(ps):
        if q is None:
            out[k * B:(k + 1) * B].zero_()
        else:
            out[k * B:(k + 1) * B].copy_(q.reshape(out[k * B:(k + 1) * B].shape))
    return 0


def gather_rows(src, idx, out, rows, words, src_rows):
    out.copy_(src.index_select(0, idx.long()).reshape(out.shape)); return 0


def add_residual(msk, out, M, C):
    m = msk.reshape(M, C)
    res = 1.0 - (m == 1).any(-1, keepdim=True).to(m.dtype)
    out.copy_(torch.cat([m, res], -1).reshape(out.shape)); return 0


def upsample2_bwd(dy, dx, B, H, W, C):
    d = dy.reshape(B, H, 2, W, 2, C).sum((2, 4))
    dx.copy_(d); return 0


def upsample2_fwd(x, y, B, H, W, C):
    y.copy_(O.upsample2(x.reshape(B, H, W, C))); return 0


def subsample_fwd(x, y, B, Ho, Wo, C, f):
    y.copy_(x.reshape(B, Ho * f, Wo * f, C)[:, ::f, ::f]); return 0


def subsample_bwd(dy, dx, B, Ho, Wo, C, f):
    d = torch.zeros(B, Ho * f, Wo * f, C, dtype=dy.dtype)
    d[:, ::f, ::f] = dy.reshape(B, Ho, Wo, C)
    dx.copy_(d); return 0


def softmax_fwd(x, p, s, npix, C):
    q = torch.softmax(x.reshape(npix, C), -1)
    p.copy_(q.reshape(p.shape))
    if s is not None:
        s.copy_(torch.round(q).reshape(s.shape))
    return 0


def softmax_bwd(dy, p, dx, npix, C):
    g, q = dy.reshape(npix, C), p.reshape(npix, C)
    dx.copy_((q * (g - (g * q).sum(-1, keepdim=True))).reshape(dx.shape)); return 0


def film_fwd(x, gamma, beta, res, y, B, HW, C, alpha):
    v = x.reshape(B, HW, C) * gamma.reshape(B, 1, C) + beta.reshape(B, 1, C)
    v = torch.where(v >= 0, v, v * alpha)
    if res is not None:
        v = v + res.reshape(B, HW, C)
    y.copy_(v.reshape(y.shape)); return 0


def film_bwd_workspace(B, C):
    return 1


def film_bwd(du, x, gamma, beta, dx, dgamma, dbeta, ws, B, HW, C, alpha):
    xr = x.reshape(B, HW, C)
    pre = xr * gamma.reshape(B, 1, C) + beta.reshape(B, 1, C)
    g = du.reshape(B, HW, C) * torch.where(pre >= 0, torch.ones_like(pre), torch.full_like(pre, alpha))
    dx.copy_((g * gamma.reshape(B, 1, C)).reshape(dx.shape))
    dgamma.copy_((g * xr).sum(1)); dbeta.copy_(g.sum(1)); return 0


def maximum_fwd(a, b, y, n):
    y.copy_(torch.maximum(a, b)); return 0


def maximum_bwd(a, b, dy, da, db, n):
    first = a >= b
    if da is not None:
        da.copy_(torch.where(first, dy, torch.zeros_like(dy)))
    if db is not None:
        db.copy_(torch.where(first, torch.zeros_like(dy), dy))
    return 0


def slice_fwd(x, y, M, C, c0, Cs):
    y.copy_(x.reshape(M, C)[:, c0:c0 + Cs].reshape(y.shape)); return 0


def slice_bwd(dy, dx, M, C, c0, Cs):
    d = torch.zeros(M, C, dtype=dy.dtype)
    d[:, c0:c0 + Cs] = dy.reshape(M, Cs)
    dx.copy_(d.reshape(dx.shape)); return 0


def sampling_kl_fwd(mu, lv, eps, z, kl, B, Z):
    z.copy_(O.sampling(mu, lv, eps)); kl.copy_(O.kl(mu, lv)); return 0


def sampling_kl_bwd(mu, lv, eps, dz, dkl, dmu, dlv, B, Z):
    gz = dz if dz is not None else torch.zeros_like(mu)
    gk = dkl.reshape(B, 1) if dkl is not None else torch.zeros(B, 1, dtype=mu.dtype)
    dmu.copy_(gz + gk * mu)
    dlv.copy_(gz * 0.5 * torch.exp(0.5 * lv) * eps + gk * (-0.5) * (1 - torch.exp(lv))); return 0


def norm_workspace_floats(C):
    return 1


def bn_stats(x, gamma, beta, mean, invstd, scale, shift, mov_mean, mov_var, ws, M, C, eps, momentum):
    xr = x.reshape(M, C)
    mu, var = xr.mean(0), xr.var(0, unbiased=False)
    istd = torch.rsqrt(var + eps)
    mean.copy_(mu); invstd.copy_(istd)
    sc = gamma * istd
    scale.copy_(sc); shift.copy_(beta - mu * sc)
    if mov_mean is not None:
        unb = var * (M / (M - 1)) if M > 1 else var
        mov_mean.sub_((mov_mean - mu) * (1 - momentum))
        mov_var.sub_((mov_var - unb) * (1 - momentum))
    return 0


def bn_infer_prep(gamma, beta, mov_mean, mov_var, scale, shift, C, eps):
    sc = gamma * torch.rsqrt(mov_var + eps)
    scale.copy_(sc); shift.copy_(beta - mov_mean * sc); return 0


def bn_apply(x, scale, shift, y, M, C, relu):
    v = x.reshape(M, C) * scale + shift
    if relu:
        v = torch.relu(v)
    y.copy_(v.reshape(y.shape)); return 0


def bn_bwd(dy, y, x, gamma, mean, invstd, dx, dgamma, dbeta, coef, ws, M, C, relu, accumulate):
    g = dy.reshape(M, C)
    if relu:
        g = g * (y.reshape(M, C) > 0).to(g.dtype)
    xh = (x.reshape(M, C) - mean) * invstd
    db, dg = g.sum(0), (g * xh).sum(0)
    if accumulate:
        dbeta.add_(db); dgamma.add_(dg)
    else:
        dbeta.copy_(db); dgamma.copy_(dg)
    dx.copy_((gamma * invstd * (g - db / M - xh * dg / M)).reshape(dx.shape)); return 0


def bn_bwd_x(dy, x, scale, shift, gamma, mean, invstd, dx, dgamma, dbeta, coef, ws, M, C, relu, accumulate):
    y = (x.reshape(M, C) * scale + shift).reshape(x.shape)          # the forward pass's pre-ReLU output (bn_apply's own expression)
    return bn_bwd(dy, y, x, gamma, mean, invstd, dx, dgamma, dbeta, coef, ws, M, C, relu, accumulate)


def in_workspace_floats(B):
    return 1


def _in_fwd(x, gamma, beta, B, per, eps, act_alpha, stat=None):
    xr = x.reshape(B, per)
    mu = xr.mean(1, keepdim=True)
    sd = xr.var(1, unbiased=False, keepdim=True).sqrt()
    rs = 1.0 / (sd + eps)
    v = (xr - mu) * rs
    if gamma is not None:
        v = v * (1 + gamma.reshape(B, per)) + beta.reshape(B, per)
    if act_alpha >= 0:
        v = torch.where(v >= 0, v, v * act_alpha)
    if stat is not None:
        stat.copy_(torch.cat([mu, rs], 1))
    return v


def instnorm_spade_fwd(x, gamma, beta, y, stat, ws, B, per, eps, act_alpha):
    y.copy_(_in_fwd(x, gamma, beta, B, per, eps, act_alpha, stat).reshape(y.shape)); return 0


def instnorm_spade_bwd(dy, x, stat, gamma, beta, dx, dgamma, dbeta, dxn, ws, B, per, eps, act_alpha):
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        ins = [xr]
        gr = br = None
        if gamma is not None:
            gr = gamma.detach().clone().requires_grad_(True)
            br = beta.detach().clone().requires_grad_(True)
            ins += [gr, br]
        out = _in_fwd(xr, gr, br, B, per, eps, act_alpha)
        gs = torch.autograd.grad(out, ins, dy.reshape(out.shape))
    dx.copy_(gs[0].reshape(dx.shape))
    if gamma is not None:
        dgamma.copy_(gs[1].reshape(dgamma.shape)); dbeta.copy_(gs[2].reshape(dbeta.shape))
    return 0


def concat_cols(a, b, out, M, Ca, Cb):
    out.reshape(M, Ca + Cb).copy_(torch.cat([a.reshape(M, Ca), b.reshape(M, Cb)], 1)); return 0


def split_cols_acc(src, da, db, M, Ca, Cb):
    s2 = src.reshape(-1)[:M * (Ca + Cb)].reshape(M, Ca + Cb)
    da.reshape(M, Ca).add_(s2[:, :Ca]); db.reshape(M, Cb).add_(s2[:, Ca:]); return 0


def instnorm_spade_fwd_gb(x, gb, y, stat, ws, B, per, C, eps, act_alpha):
    g2 = gb.reshape(-1, 2 * C)
    return instnorm_spade_fwd(x, g2[:, :C].contiguous(), g2[:, C:].contiguous(), y, stat, ws, B, per, eps, act_alpha)


def instnorm_spade_bwd_gb(dy, x, stat, gb, dx, dgb, dxn, ws, B, per, C, eps, act_alpha):
    g2 = gb.reshape(-1, 2 * C)
    dg, db = torch.empty_like(x), torch.empty_like(x)
    rc = instnorm_spade_bwd(dy, x, stat, g2[:, :C].contiguous(), g2[:, C:].contiguous(), dx, dg, db, dxn, ws, B, per, eps, act_alpha)
    dgb.reshape(-1, 2 * C).copy_(torch.cat([dg.reshape(-1, C), db.reshape(-1, C)], 1))
    return rc


def instnorm_spade_fwd_gb_t(x, gb, y, stat, ws, B, per, C, eps, act_alpha, h, hy):
    tmp = torch.empty(y.shape, dtype=torch.float32)
    rc = instnorm_spade_fwd_gb(x, gb.float(), tmp, stat, ws, B, per, C, eps, act_alpha)
    y.copy_(tmp.to(y.dtype))
    return rc


def instnorm_spade_bwd_gb_t(dy, x, stat, gb, dx, dgb, dxn, ws, B, per, C, eps, act_alpha, h, hy):
    tmp = torch.empty(dgb.shape, dtype=torch.float32)
    rc = instnorm_spade_bwd_gb(dy.float(), x, stat, gb.float(), dx, tmp, dxn, ws, B, per, C, eps, act_alpha)
    dgb.copy_(tmp.to(dgb.dtype))
    return rc


def colsum_t(x, out, ws, M, C, accumulate, h):
    s = x.float().reshape(M, C).sum(0)
    out.copy_(out + s if accumulate else s); return 0


def dense_workspace_floats(R, K, N):
    return 1


def dense_fwd(x, w, bias, y, ws, R, K, N, act, alpha):
    v = x.reshape(R, K) @ w.reshape(K, N)
    if bias is not None:
        v = v + bias
    y.copy_(_act(v, act, alpha)); return 0


def dense_dgrad(dy, w, dx, R, K, N):
    dx.copy_(dy.reshape(R, N) @ w.reshape(K, N).t()); return 0


def dense_wgrad(x, dy, dw, R, K, N, accumulate):
    v = (x.reshape(R, K).t() @ dy.reshape(R, N)).reshape(dw.shape)
    if accumulate:
        dw.add_(v)
    else:
        dw.copy_(v)
    return 0


def tps_workspace_floats(B):
    return 1


def _tps_loc(theta, Mb, B, H, W):
    q = O.nd_grid((H, W), theta.dtype)[0]                  # [HW, 2] (row, col)
    ln = q[None] + torch.einsum('pj,bjk->bpk', Mb, theta.reshape(B, 25, 2))
    return torch.flip(ln, dims=[-1]) * torch.tensor([W - 1, H - 1], dtype=theta.dtype)


def tps_warp_fwd(vol, theta, Mb, out, loc, B, H, W, C):
    l = _tps_loc(theta, Mb, B, H, W)
    if loc is not None:
        loc.copy_(l.reshape(loc.shape))
    out.copy_(O.resampler(vol.reshape(B, H, W, C), l).reshape(out.shape)); return 0


def tps_scatter_workspace_floats(B, H, W, C):
    return 2


def tps_warp_bwd(vol, loc, Mb, dout, dvol, dtheta, dloc, ws, acc, B, H, W, C):
    with torch.enable_grad():
        v = vol.detach().clone().reshape(B, H, W, C).requires_grad_(True)
        l = loc.detach().clone().reshape(B, H * W, 2).requires_grad_(True)
        o = O.resampler(v, l)
        gv, gl = torch.autograd.grad(o, [v, l], dout.reshape(o.shape))
    if dvol is not None:
        dvol.copy_(gv.reshape(dvol.shape))
    if dtheta is not None:
        glr = torch.flip(gl * torch.tensor([W - 1, H - 1], dtype=gl.dtype), dims=[-1])   # -> (row, col) normalised
        dtheta.copy_(torch.einsum('pj,bpk->bjk', Mb, glr).reshape(dtheta.shape))
    return 0


def segloss_workspace_floats(B):
    return 1


def segloss_stats_floats(B):
    return 3 * B + 16


def segloss_coef_floats(B, C):
    return 2 * B + 2 * C


def segloss_class_offset(B):
    return 3 * B


def segloss_stats(pred, target, stats, ws, B, HW, C, nm):
    p, t = pred.reshape(B, HW, C), target.reshape(B, HW, C)
    st = torch.zeros(3 * B + 16, dtype=pred.dtype)
    st[0:3 * B:3] = (t[..., :nm] * p[..., :nm]).sum((1, 2))
    st[1:3 * B:3] = t[..., :nm].sum((1, 2))
    st[2:3 * B:3] = p[..., :nm].sum((1, 2))
    st[3 * B:3 * B + C] = p.sum((0, 1))
    st[3 * B + 8:3 * B + 8 + C] = (p * torch.log(t + 1e-12)).sum((0, 1))
    stats.copy_(st); return 0


def segloss_finalize(stats, loss, coef, B, C, n_pix_global, n_pix_grad, lambda_bce):
    I, T, P = stats[0:3 * B:3], stats[1:3 * B:3], stats[2:3 * B:3]
    den, num = T + P + 1e-12, 2 * I + 1e-12
    dice = (1 - num / den).mean()
    cf = torch.zeros(2 * B + 2 * C, dtype=stats.dtype)
    cf[0:2 * B:2] = -2.0 / den / B
    cf[1:2 * B:2] = num / (den * den) / B
    bce = torch.zeros((), dtype=stats.dtype)
    if lambda_bce != 0:
        n, S = stats[3 * B:3 * B + C], stats[3 * B + 8:3 * B + 8 + C]
        Tt = n.sum()
        w = Tt / (n + 1e-12)
        bce = -(w * S).sum() / n_pix_global
        sumq = (S / (n + 1e-12)).sum()
        cf[2 * B:2 * B + C] = -lambda_bce / n_pix_grad * w
        cf[2 * B + C:2 * B + 2 * C] = -lambda_bce / n_pix_grad * (sumq - S * Tt / (n + 1e-12) ** 2)
    loss.copy_((dice + lambda_bce * bce).reshape(1)); coef.copy_(cf); return 0


def segloss_grad(pred, target, coef, dpred, B, HW, C, nm, scale, use_bce):
    t = target.reshape(B, HW, C)
    g = torch.zeros_like(t)
    g[..., :nm] = coef[0:2 * B:2].reshape(B, 1, 1) * t[..., :nm] + coef[1:2 * B:2].reshape(B, 1, 1)
    if use_bce:
        g = g + coef[2 * B:2 * B + C] * torch.log(t + 1e-12) + coef[2 * B + C:2 * B + 2 * C]
    dpred.copy_((g * scale).reshape(dpred.shape)); return 0


def diffloss_workspace_floats():
    return 1


def diffloss(p, t, tconst, n, mode, loss, ws):
    d = p - (t if t is not None else tconst)
    v = d.abs().mean() if mode == 0 else ((d * d).mean() if mode == 1 else p.mean())
    loss.copy_(v.reshape(1)); return 0


def diffloss_grad(p, t, tconst, n, mode, scale, dp):
    d = p - (t if t is not None else tconst)
    g = torch.sign(d) if mode == 0 else (2 * d if mode == 1 else torch.ones_like(d))
    dp.copy_(g * scale); return 0


def adam(p, g, m, v, n, lr_t, b1, b2, eps):
    m.mul_(b1).add_(g * (1 - b1))
    v.mul_(b2).add_(g * g * (1 - b2))
    p.sub_(lr_t * m / (v.sqrt() + eps)); return 0


def spectral_workspace_floats(K, N):
    return 1


def spectral_fwd(w, u0, loss, sgn, ws, K, N, alpha):
    x = w.reshape(K, N)
    u = u0.reshape(K, 1)
    for _ in range(3):
        wtu = x.t() @ u
        v = wtu / wtu.norm()
        wv = x @ v
        u = wv / wv.norm()
    sigma = wv.norm()
    d = 1 - 1 / sigma
    loss.copy_((alpha * d.abs() * x.abs().mean()).reshape(1))
    sgn.copy_((alpha / x.numel() * torch.sign(d)).reshape(1)); return 0


def spectral_fwd4(w0, w1, w2, w3, u0, u1, u2, u3, loss, sgn, ws, n, K0, N0, K1, N1, K2, N2, K3, N3, alpha):
    for i, (w, u, K, N_) in enumerate(((w0, u0, K0, N0), (w1, u1, K1, N1), (w2, u2, K2, N2), (w3, u3, K3, N3))[:n]):
        spectral_fwd(w, u, loss[i:i + 1], sgn[i:i + 1], None, K, N_, alpha)
    return 0


def spectral_grad4(w0, w1, w2, w3, sgn, d0, d1, d2, d3, n, n0, n1, n2, n3, scale):
    for i, (w, d) in enumerate(((w0, d0), (w1, d1), (w2, d2), (w3, d3))[:n]):
        d.add_(torch.sign(w) * sgn[i] * scale)
    return 0


def spectral_grad(w, sgn, scale, n, dw):
    dw.copy_(torch.sign(w) * sgn * scale); return 0


def pairloss_workspace_floats(B):
    return 16


def pair_dice_fwd(a, b, stats, out, ldo, ws, B, per):
    a2, b2 = a.reshape(-1)[:B * per].reshape(B, per), b.reshape(-1)[:B * per].reshape(B, per)
    I, A, Bs = (a2 * b2).sum(1), a2.sum(1), b2.sum(1)
    stats.reshape(-1)[:3 * B].copy_(torch.stack([I, A, Bs], 1).reshape(-1))
    d = (2 * I + 1e-12) / (A + Bs + 1e-12)
    out[:(B - 1) * ldo + 1:ldo].copy_(d)
    return 0


def pair_dice_bwd(a, b, stats, g, ldg, da, db, acc, B, per):
    a2, b2 = a.reshape(B, per), b.reshape(B, per)
    st = stats.reshape(-1)[:3 * B].reshape(B, 3)
    den, num = st[:, 1] + st[:, 2] + 1e-12, 2 * st[:, 0] + 1e-12
    gs = g[:(B - 1) * ldg + 1:ldg]
    k1, k0 = (gs * 2 / den)[:, None], (-gs * num / den ** 2)[:, None]
    ga = k1 * b2 + k0
    if acc:
        da.reshape(B, per).add_(ga)
    else:
        da.reshape(B, per).copy_(ga)
    if db is not None:
        db.reshape(B, per).copy_(k1 * a2 + k0)
    return 0


def row_mae_fwd(x, y, out, ws, B, per):
    out.copy_((x.reshape(B, per) - y.reshape(B, per)).abs().mean(1))
    return 0


def row_mae_bwd(x, y, g, dy, B, per):
    dy.reshape(B, per).copy_(torch.sign(y.reshape(B, per) - x.reshape(B, per)) * (g.reshape(B, 1) / per))
    return 0


def segpb_stats_floats(B):
    return B * 11 + 8


def segpb_class_offset(B):
    return B * 11


def _label_nll(t):
    return -torch.log(torch.softmax(t, -1) + 1e-12)


def segpb_stats(pred, target, stats, ws, B, HW, C, nm):
    p, t = pred.reshape(B, HW, C), target.reshape(B, HW, C)
    st = torch.zeros(B, 11)
    st[:, 0] = (t[..., :nm] * p[..., :nm]).sum((1, 2))
    st[:, 1] = t[..., :nm].sum((1, 2))
    st[:, 2] = p[..., :nm].sum((1, 2))
    st[:, 3:3 + C] = (p * _label_nll(t)).sum(1)
    n = torch.zeros(8)
    n[:C] = p.sum((0, 1))
    stats.copy_(torch.cat([st.reshape(-1), n]))
    return 0


def segpb_loss(stats, loss, B, HW, C, lam):
    st, n = stats[:B * 11].reshape(B, 11), stats[B * 11:B * 11 + C]
    w = n.sum() / (n + 1e-12)
    loss.copy_(1 - (2 * st[:, 0] + 1e-12) / (st[:, 1] + st[:, 2] + 1e-12) + lam / HW * (st[:, 3:3 + C] * w).sum(1))
    return 0


def segpb_classgrad(stats, g, A, B):
    st = stats[:B * 11].reshape(B, 11)
    A.copy_((g.reshape(B, 1) * st[:, 3:]).sum(0))
    return 0


def segpb_grad(target, stats, g, A, dpred, B, HW, C, nm, lam):
    t = target.reshape(B, HW, C)
    st, n = stats[:B * 11].reshape(B, 11), stats[B * 11:B * 11 + C]
    N_ = n.sum()
    d = n + 1e-12
    w = N_ / d
    K = (A[:C] / d).sum() - A[:C] * N_ / d ** 2
    gb = g.reshape(B, 1, 1)
    out = lam / HW * (gb * _label_nll(t) * w + K)
    den, num = (st[:, 1] + st[:, 2] + 1e-12).reshape(B, 1, 1), (2 * st[:, 0] + 1e-12).reshape(B, 1, 1)
    out[..., :nm] += gb * (-2 * t[..., :nm] / den + num / den ** 2)
    dpred.copy_(out.reshape(dpred.shape))
    return 0


def rowdot_fwd(w, l, out, B, J):
    out.copy_((w * l).sum(1, keepdim=True))
    return 0


def rowdot_bwd(w, l, g, dw, dl, B, J):
    dw.copy_(g.reshape(B, 1) * l)
    dl.copy_(g.reshape(B, 1) * w)
    return 0


def affine_gather(data, rows, mat, out, B, H, W, C, order):
    src = data if rows is None else data.index_select(0, rows.long())
    r = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    c = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    m = mat.view(B, 6, 1, 1)
    sr = (m[:, 0] * r + m[:, 1] * c + m[:, 2]).clamp(0, H - 1)
    sc = (m[:, 3] * r + m[:, 4] * c + m[:, 5]).clamp(0, W - 1)
    if order == 0:
        sr, sc = torch.floor(sr + 0.5), torch.floor(sc + 0.5)
    r0, c0 = sr.floor().long().clamp(max=H - 1), sc.floor().long().clamp(max=W - 1)
    r1, c1 = (r0 + 1).clamp(max=H - 1), (c0 + 1).clamp(max=W - 1)
    ar, ac = (sr - r0).unsqueeze(-1), (sc - c0).unsqueeze(-1)
    bi = torch.arange(B).view(B, 1, 1)
    g = lambda rr, cc: src[bi, rr, cc]
    out.copy_((1 - ar) * ((1 - ac) * g(r0, c0) + ac * g(r0, c1)) + ar * ((1 - ac) * g(r1, c0) + ac * g(r1, c1)))
    return 0


_TABLE = {('mmseg_' + k): v for k, v in list(globals().items()) if callable(v) and not k.startswith('_')
          and k not in ('install', 'uninstall')}


def _call(name, *args):
    fn = _TABLE.get(name)
    if fn is None:
        raise KeyError('cpu_backend has no stand-in for %s' % name)
    with torch.no_grad():
        return fn(*args)


_saved = {}


def install():
    from multimodal_segmentation_amd import _native
    if 'call' not in _saved:
        _saved['call'] = _native.call
    _native.call = _call


def uninstall():
    from multimodal_segmentation_amd import _native
    if 'call' in _saved:
        _native.call = _saved.pop('call')
