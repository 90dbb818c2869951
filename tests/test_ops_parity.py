"""Op-level parity: every operator of multimodal_segmentation_amd.ops (forward and backward through the C ABI)
against the oracle on the same seeded inputs.

  * `-m gpu`     : the real libmmseg_hip.so on cuda:0 (the parity tests proper);
  * `-m "not gpu"`: the same cases through tests/cpu_backend.py, which checks the host-side autograd glue.
Tolerances: fp32 accumulation over K <= 4608 terms -> 2e-4 relative to the tensor's max magnitude.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from multimodal_segmentation_amd import ops as P

RTOL = 2e-4


def _dev(request):
    return request.param


def _close(a, b, name, rtol=RTOL):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, '%s: shape %s vs %s' % (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + 1e-7, '%s: max err %.3e (scale %.3e)' % (name, err, scale)


def check(f_prod, f_ref, inputs, device, grad_mask=None, rtol=RTOL, param_idx=()):
    """inputs: list of CPU fp32 tensors.  Runs both functions, a random cotangent, and compares outputs and the
    gradients of every input with grad_mask[i] true.  Inputs listed in `param_idx` are weights: the product gets
    them as plain tensors plus a zeroed gradient buffer `t.gbuf` it accumulates into (no autograd leaf)."""
    grad_mask = grad_mask or [True] * len(inputs)
    xs_p = []
    for i, (t, m) in enumerate(zip(inputs, grad_mask)):
        tp = t.clone().to(device)
        if i in param_idx:
            tp.gbuf = torch.zeros_like(tp)
        else:
            tp.requires_grad_(m)
        xs_p.append(tp)
    xs_r = [t.clone().double().requires_grad_(m) for t, m in zip(inputs, grad_mask)]
    yp = f_prod(*xs_p)
    yr = f_ref(*xs_r)
    yp = yp if isinstance(yp, (tuple, list)) else [yp]
    yr = yr if isinstance(yr, (tuple, list)) else [yr]
    g = torch.Generator().manual_seed(123)
    cots = [torch.randn(y.shape, generator=g) for y in yr]
    pre = getattr(f_ref, 'pre', None)
    if pre is not None:   # no cotangent where the activation kink could flip between fp32 and fp64
        cots[0] = cots[0] * (pre.detach().abs() > 1e-4).float()
    for i, (a, b) in enumerate(zip(yp, yr)):
        _close(a, b, 'out%d' % i, rtol)
    torch.autograd.backward(list(yp), [c.to(device) for c in cots])
    torch.autograd.backward(list(yr), [c.double() for c in cots])
    for i, m in enumerate(grad_mask):
        if m:
            gp = xs_p[i].gbuf if i in param_idx else xs_p[i].grad
            assert gp is not None, 'no grad for input %d' % i
            _close(gp, xs_r[i].grad, 'grad%d' % i, rtol)


def _anchor(t):
    return torch.zeros(1, device=t.device, requires_grad=True)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


DEVICES = [pytest.param('cpu', id='cpu-standin'), pytest.param('cuda', marks=pytest.mark.gpu, id='mi355x')]


@pytest.fixture(params=DEVICES)
def device(request):
    if request.param == 'cpu':
        from tests import cpu_backend as cb
        cb.install()
        yield 'cpu'
        cb.uninstall()
    else:
        assert torch.cuda.is_available(), 'gpu-marked test needs a GPU'
        yield 'cuda'


# B, H, W, C1, C2, Cout, k, stride, padding, act, ups
CONV_CASES = [
    (2, 16, 16, 64, 0, 64, 3, 1, 'same', None, False),
    (2, 20, 12, 1, 0, 64, 3, 1, 'same', 'relu', False),
    (2, 16, 16, 8, 0, 8, 3, 1, 'same', 'leaky', False),
    (1, 16, 16, 64, 64, 64, 3, 1, 'same', None, False),
    (2, 16, 16, 128, 0, 64, 3, 1, 'same', None, True),
    (2, 33, 33, 8, 1, 16, 3, 2, 'valid', 'leaky', False),
    (2, 34, 30, 4, 0, 64, 4, 2, 'valid', 'leaky', False),
    (2, 16, 16, 64, 0, 128, 4, 2, 'valid', 'leaky', False),
    (2, 9, 9, 64, 0, 128, 4, 1, 'valid', 'leaky', False),
    (2, 20, 20, 8, 8, 20, 5, 1, 'valid', 'leaky', False),
    (3, 9, 12, 8, 8, 20, 5, 1, 'valid', 'leaky', False),       # locnet first layer (s2conv.hpp): less than one tile, batch 3
    (1, 40, 70, 8, 8, 20, 5, 1, 'valid', None, False),         # ... several tiles, the last column tile partly outside the image
    (2, 18, 22, 20, 0, 20, 5, 1, 'valid', 'leaky', False),     # locnet second / third layer and its data gradient (padding 4, flipped kernel)
    (1, 13, 80, 20, 0, 20, 5, 1, 'valid', 'leaky', False),     # ... two column tiles, two row tiles
    (2, 16, 16, 64, 0, 8, 1, 1, 'same', None, False),
    (2, 16, 16, 64, 0, 5, 1, 1, 'same', None, False),
    (2, 16, 16, 8, 0, 1, 1, 1, 'same', 'tanh', False),
    (1, 12, 12, 1, 0, 64, 4, 2, 'valid', 'leaky', False),
    (2, 8, 8, 256, 0, 512, 3, 1, 'same', None, False),
    (3, 128, 128, 64, 0, 128, 3, 1, 'same', 'relu', False),
    (2, 128, 128, 8, 0, 8, 3, 1, 'same', 'leaky', False),      # large M, tiny C: folded column-sum path
    (3, 16, 192, 8, 0, 8, 3, 1, 'same', None, False),          # ... three tiles per row band, batch 3 (weight gradient on the 4x4x1 MFMA)
    (2, 96, 96, 64, 0, 5, 1, 1, 'same', None, False),          # large M, C not dividing 64: generic column sums
    (2, 33, 31, 1, 0, 64, 4, 2, 'valid', 'leaky', False),      # D_Image first layer, odd sizes: direct small-Cin data gradient
    (2, 33, 33, 16, 0, 32, 3, 2, 'valid', 'leaky', False),     # modality-encoder layer: batched parity classes with 2x1 / 1x1 taps
    (2, 31, 31, 64, 0, 128, 4, 2, 'valid', 'leaky', False),    # odd input: parity classes of different sizes in one launch
    (2, 18, 22, 8, 0, 64, 3, 1, 'same', 'relu', False),        # segmentor c0: its data gradient is a 64 -> 8 3x3 launch (N = 8)
    (3, 17, 19, 64, 0, 8, 3, 1, 'same', 'leaky', False),       # N = 8 on the 32-wide fast tile, odd sizes
    (3, 17, 19, 8, 0, 5, 1, 1, 'same', None, False),           # tiny 1x1 head through the generic kernel
    (2, 20, 21, 8, 0, 64, 3, 1, 'valid', 'leaky', False),      # few input channels, 'valid': data gradient = 1x1 GEMM over the taps + tap sum
    (2, 16, 20, 4, 0, 32, 3, 1, 'same', None, False),          # Cin = 4 through the same path
    (1, 24, 24, 16, 0, 128, 3, 1, 'same', 'relu', False),      # Cin = 16: 144 GEMM columns
    # round 3: the HBM-bound small-channel kernels (csrc/smallconv.hpp), pixel counts that do not fill the last wave / block
    (3, 17, 19, 64, 0, 5, 1, 1, 'same', None, False),          # segmentor head: pw_reduce<16, 5> + its wgrad; dgrad = smallk<1, 5>
    (3, 17, 19, 64, 0, 8, 1, 1, 'same', None, False),          # anatomy head: pw_reduce<16, 8>; dgrad = smallk<1, 8>
    (5, 13, 11, 8, 0, 1, 1, 1, 'same', 'tanh', False),         # FiLM decoder head: pw_reduce<2, 1>; dgrad = smallk<1, 1> (Cout 8)
    (3, 15, 9, 16, 0, 1, 1, 1, 'same', 'tanh', False),         # SPADE decoder head: pw_reduce<4, 1>
    (3, 17, 19, 1, 0, 64, 3, 1, 'same', 'relu', False),        # UNet d_l0.a: smallk<3, 1> forward + weight gradient
    (2, 37, 41, 1, 0, 64, 4, 2, 'valid', 'leaky', False),      # D_Image first layer: smallk<4, 1>
    (2, 19, 23, 1, 0, 16, 3, 1, 'same', None, False),          # 4 lanes per pixel
    (1, 70, 70, 64, 0, 5, 1, 1, 'same', None, False),          # more pixels than one pass of the grid
    # the discriminators' 4x4 layers at odd output widths: weight gradient on the any-width transposed-staging kernel (pixel quads that
    # straddle row and sample boundaries, stride 2 and stride 1, a pixel count that is not a multiple of 4)
    (3, 64, 60, 64, 0, 128, 4, 2, 'valid', 'leaky', False),    # Ho x Wo = 31 x 29
    (3, 30, 30, 64, 0, 256, 4, 1, 'valid', 'leaky', False),    # 27 x 27: M = 2187
    # round 4: the modality encoder's first layer on its own kernels (csrc/s2conv.hpp): parity classes of different sizes, pixel counts
    # that fill neither the last 16-pixel tile nor the last group of 4, an even height whose last row no tap reaches
    (3, 20, 37, 8, 1, 16, 3, 2, 'valid', 'leaky', False),
    (1, 64, 64, 8, 1, 16, 3, 2, 'valid', None, False),
    (5, 7, 9, 8, 1, 16, 3, 2, 'valid', 'leaky', False),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d(case, device):
    B, H, W, C1, C2, Cout, k, stride, padding, act, ups = case
    alpha = 0.2 if act == 'leaky' else 0.0
    x1 = rnd(B, H // 2 if ups else H, W // 2 if ups else W, C1, seed=1)
    w = rnd(k, k, C1 + C2, Cout, seed=2, scale=(2.0 / (k * k * (C1 + C2))) ** 0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    inputs = [x1, w, b] + ([rnd(B, H, W, C2, seed=4)] if C2 else [])

    def f_prod(x1, w, b, x2=None):
        return P.conv2d(x1, w, b, stride=stride, padding=padding, act=act, alpha=alpha, x2=x2, upsample=ups,
                        wgrad=w.gbuf, bgrad=b.gbuf, anchor=_anchor(x1))

    def f_ref(x1, w, b, x2=None):
        xin = O.upsample2(x1) if ups else x1
        if x2 is not None:
            xin = torch.cat([xin, x2], -1)
        y = O.conv2d(xin, w, b, stride=stride, padding=padding)
        f_ref.pre = y
        if act == 'relu':
            y = torch.relu(y)
        elif act == 'leaky':
            y = O.leaky_relu(y, alpha)
        elif act == 'tanh':
            y = torch.tanh(y)
        return y

    check(f_prod, f_ref, inputs, device, param_idx=(1, 2))


@pytest.mark.gpu
def test_modality_encoder_first_layer_runs_on_its_own_kernels():
    """Conv2D(16, 3, strides=2, 'valid') over Concatenate([anatomy 8, image 1]) (modality_encoder.py:34-38 of the reference): forward, weight
    gradient and data gradient leave the generic implicit-GEMM kernels (csrc/s2conv.hpp; values: the CONV_CASES above against the oracle)"""
    from multimodal_segmentation_amd import _native as N
    dev = 'cuda'
    x1 = rnd(2, 33, 35, 8, seed=1).to(dev)
    x2 = rnd(2, 33, 35, 1, seed=2).to(dev)
    w = rnd(3, 3, 9, 16, seed=3, scale=0.2).to(dev)
    b = rnd(16, seed=4, scale=0.1).to(dev)
    for need_dx in (False, True):
        xa = x1.clone().requires_grad_(need_dx)
        wg, bg = torch.zeros_like(w), torch.zeros_like(b)
        anchor = torch.zeros(1, device=dev, requires_grad=True)
        y = P.conv2d(xa, w, b, stride=2, padding='valid', act='leaky', alpha=0.2, x2=x2, wgrad=wg, bgrad=bg, anchor=anchor)
        assert N.call('mmseg_conv2d_last_kernel') == 21016016
        y.backward(torch.ones_like(y))
        # the backward node queues the weight gradient, then (if an input wants it) the data gradient
        assert N.call('mmseg_conv2d_last_kernel') == (21016009 if need_dx else 22016016)
        assert float(wg.abs().sum()) > 0


@pytest.mark.parametrize('B,H,W,Cin,f,alpha', [(2, 12, 10, 128, 32, 0.2), (2, 9, 11, 128, 16, -1.0), (1, 16, 16, 128, 128, 0.2),
                                               (2, 8, 8, 64, 64, 0.2)])
def test_spade_gamma_beta_as_one_convolution(B, H, W, Cin, f, alpha, device):
    """ops.conv2d_pair + ops.instnorm_spade_gb (round 3: the gamma and beta convolutions of a SPADE unit as ONE convolution with 2f output
    channels, InstanceNorm + modulation reading the halves of its output) against the oracle's two separate convolutions
    (layers/spade.py:26-33 of the reference): outputs, the gradients of both inputs and of the four parameters"""
    a = rnd(B, H, W, Cin, seed=1)
    x = rnd(B, H, W, f, seed=2) * 1.5 + 0.3
    wg = rnd(3, 3, Cin, f, seed=3, scale=(2.0 / (9 * Cin)) ** 0.5)
    wb = rnd(3, 3, Cin, f, seed=4, scale=(2.0 / (9 * Cin)) ** 0.5)
    bg, bb = rnd(f, seed=5, scale=0.1), rnd(f, seed=6, scale=0.1)

    def f_prod(a, x, wg, bg, wb, bb):
        gb = P.conv2d_pair(a, wg, bg, wb, bb, (wg.gbuf, bg.gbuf, wb.gbuf, bb.gbuf), anchor=_anchor(a), wkey=None)
        return P.instnorm_spade_gb(x, gb, alpha)

    def f_ref(a, x, wg, bg, wb, bb):
        gamma, beta = O.conv2d(a, wg, bg), O.conv2d(a, wb, bb)
        u = O.instance_norm(x) * (1 + gamma) + beta
        f_ref.pre = u if alpha >= 0 else None
        return O.leaky_relu(u, alpha) if alpha >= 0 else u

    check(f_prod, f_ref, [a, x, wg, bg, wb, bb], device, param_idx=(2, 3, 4, 5))


@pytest.mark.parametrize('n', [2, 7, 15])
def test_shared_tensor_gradients_are_summed_by_the_library(n, device):
    """ops.Shared / ops.share: n consumers of one tensor, each through its own alias -> the gradient is the sum of the n cotangents
    (one mmseg_sum_n_t launch per 8 operands), no addition is left to the autograd engine"""
    x = rnd(2, 6, 5, 8, seed=3).to(device).requires_grad_(True)
    ws = [rnd(2, 6, 5, 8, seed=10 + i) for i in range(n)]
    sh = P.Shared(x, n)
    total = None
    for i in range(n):
        a = sh.use()
        assert a.data_ptr() == x.data_ptr()
        t = P.axpby(a, a, 0.5, 0.5)      # a library op without a tape entry: keep the alias itself as the graph output
        del t
    with pytest.raises(RuntimeError):
        sh.use()
    outs = list(P.share(x, n))
    torch.autograd.backward(outs, [w.to(device) for w in ws])
    _close(x.grad, sum(w.double() for w in ws), 'shared grad')
    # unused aliases contribute nothing (and are not materialised as zeros)
    x.grad = None
    outs = list(P.share(x, n))
    torch.autograd.backward(outs[:1], [ws[0].to(device)])
    _close(x.grad, ws[0].double(), 'one consumer')


@pytest.mark.parametrize('n,shape', [(2, (3, 4, 4, 8)), (6, (2, 8)), (15, (1, 3, 3, 4))])
def test_cat_and_split_batch(n, shape, device):
    """ops.cat_batch / ops.split_batch (mmseg_cat_words): values, and gradients through both directions incl. unused splits"""
    parts = [rnd(*shape, seed=20 + i) for i in range(n)]
    xs = [t.clone().to(device).requires_grad_(True) for t in parts]
    y = P.cat_batch(xs)
    _close(y, torch.cat(parts, 0).double(), 'cat')
    cot = rnd(*y.shape, seed=5)
    y.backward(cot.to(device))
    for i, x in enumerate(xs):
        _close(x.grad, cot[i * shape[0]:(i + 1) * shape[0]].double(), 'cat grad %d' % i)
    big = rnd(n * shape[0], *shape[1:], seed=6).to(device).requires_grad_(True)
    sp = P.split_batch(big, n)
    for i, t in enumerate(sp):
        _close(t, big.detach().cpu()[i * shape[0]:(i + 1) * shape[0]].double(), 'split %d' % i)
    use = [i for i in range(n) if i % 2 == 0]
    cots = {i: rnd(*shape, seed=40 + i) for i in use}
    torch.autograd.backward([sp[i] for i in use], [cots[i].to(device) for i in use])
    want = torch.zeros(n * shape[0], *shape[1:], dtype=torch.float64)
    for i in use:
        want[i * shape[0]:(i + 1) * shape[0]] = cots[i].double()
    _close(big.grad, want, 'split grad')


def test_maxpool_with_skip(device):
    """ops.maxpool2_skip: pooled output + the skip alias; the two gradients meet in ONE pass (mmseg_maxpool2_bwd_add_t)"""
    x0 = rnd(2, 8, 6, 8, seed=2)
    x = x0.clone().to(device).requires_grad_(True)
    y, skip = P.maxpool2_skip(x)
    xr = x0.clone().double().requires_grad_(True)
    yr = O.maxpool2(xr)
    _close(y, yr, 'pooled')
    assert skip.data_ptr() == x.data_ptr()
    gy, gs = rnd(*yr.shape, seed=3), rnd(*x0.shape, seed=4)
    torch.autograd.backward([y, skip], [gy.to(device), gs.to(device)])
    torch.autograd.backward([yr, xr * 1.0], [gy.double(), gs.double()])
    _close(x.grad, xr.grad, 'pool + skip grad')
    x.grad = None
    y, skip = P.maxpool2_skip(x)
    y.backward(gy.to(device))                  # skip unused
    xr.grad = None
    O.maxpool2(xr).backward(gy.double())
    _close(x.grad, xr.grad, 'pool grad only')
    x.grad = None
    y, skip = P.maxpool2_skip(x)
    skip.backward(gs.to(device))               # pooling unused
    _close(x.grad, gs.double(), 'skip grad only')


def test_gather_rows_and_add_residual(device):
    pool = rnd(12, 4, 4, 3, seed=1).to(device)
    idx = torch.tensor([5, 0, 11, 5, 7], dtype=torch.int64).to(device)
    got = P.gather_rows(pool, idx)
    assert torch.equal(got.cpu(), pool.cpu().index_select(0, idx.cpu()))
    m = (torch.rand(3, 5, 5, 4, generator=torch.Generator().manual_seed(3)) > 0.8).float()
    m[0, 0, 0] = torch.tensor([0.5, 0.999, 0., 0.])      # fractional edge values count as background
    got = P.add_residual(m.to(device)).cpu()
    res = 1.0 - (m == 1).any(-1, keepdim=True).float()
    assert torch.equal(got, torch.cat([m, res], -1))


def test_anonymous_pair_operands_are_never_cached(device):
    """regression (round-3 driver run): ops.conv2d_pair with wkey=None cached the concatenated gamma|beta operands under the ADDRESS of
    the first kernel.  Free that kernel, allocate a DIFFERENT one of the same byte size (the caching allocator hands out the same block)
    and call again: the result must be the new weights' -- also when every shape matches, where the stale entry used to give wrong
    numbers without any error -- and a kernel of another geometry in the same block must not trip over the old operands."""
    B, H, W, Cin, f = 1, 8, 8, 64, 64
    a = rnd(B, H, W, Cin, seed=1).to(device)

    def run(seed, Cin_, f_):
        x = a if Cin_ == Cin else rnd(B, H, W, Cin_, seed=9).to(device)
        wg = rnd(3, 3, Cin_, f_, seed=seed, scale=0.05).to(device)
        wb = rnd(3, 3, Cin_, f_, seed=seed + 1, scale=0.05).to(device)
        bg, bb = rnd(f_, seed=seed + 2).to(device), rnd(f_, seed=seed + 3).to(device)
        ptr = wg.data_ptr()
        with torch.no_grad():
            y = P.conv2d_pair(x, wg, bg, wb, bb, wkey=None)
        ref = torch.cat([O.conv2d(x.cpu().double(), wg.cpu().double(), bg.cpu().double()),
                         O.conv2d(x.cpu().double(), wb.cpu().double(), bb.cpu().double())], -1)
        _close(y, ref, 'pair(seed %d)' % seed)
        out = y.cpu()
        del wg, wb, bg, bb, y
        return out, ptr

    y1, p1 = run(10, Cin, f)
    y2, p2 = run(20, Cin, f)                   # same byte size, other values
    assert (y1 - y2).abs().max().item() > 1e-2, 'second call returned the first weights\' result'
    run(30, 128, 32)                           # 3*3*128*32 floats = the same 147 456 bytes, other geometry (the driver's failing pair)
    if device == 'cuda':
        assert p1 == p2, 'allocator did not recycle the block: the regression is not exercised'
    assert not P._pair_cache, 'anonymous operands must not enter the cache'


def test_model_caches_are_evicted_with_the_model(device):
    """cached weight images / fused operands are keyed on never-reused serial numbers and leave HBM when their model dies"""
    import gc
    from multimodal_segmentation_amd import nn
    nn.set_default_device(device)
    try:
        def build_and_run():
            m = nn.Model('evict_probe')
            nn.conv_params(m, 'g', 3, 64, 64)
            nn.conv_params(m, 'b', 3, 64, 64)
            nn.conv_params(m, 'c', 3, 64, 64)
            m.finalize(np.random.RandomState(0), torch.device(device))
            x = rnd(1, 8, 8, 64).to(device)
            with torch.no_grad():
                nn.conv_pair(m, 'g', 'b', x)
                nn.conv(m, 'c', x)
            return m.uid
        before = (len(P._wprep_cache), len(P._pair_cache))
        uid = build_and_run()
        gc.collect()
        owners = [P._key_owner(k) for cache in (P._wprep_cache, P._pair_cache, P._bnfold_cache) for k in cache]
        assert uid not in owners and uid not in P._wprep_batch and uid not in P._owner_version
        assert (len(P._wprep_cache), len(P._pair_cache)) == before
        uid2 = build_and_run()
        assert uid2 != uid
    finally:
        nn.set_default_device('cuda' if torch.cuda.is_available() else 'cpu')


@pytest.mark.parametrize('shape,relu', [((2, 16, 16, 64), True), ((3, 8, 8, 128), False), ((2, 32, 32, 64), True)])
def test_batchnorm_train(shape, relu, device):
    C = shape[-1]
    x = rnd(*shape, seed=5) * 2 + 3.0          # non-zero mean: exercises the shifted-sum variance
    gamma, beta = rnd(C, seed=6) * 0.2 + 1, rnd(C, seed=7) * 0.2
    mm0, mv0 = rnd(C, seed=8) * 0.1, torch.rand(C) + 0.5
    mm_p, mv_p = mm0.clone().to(device), mv0.clone().to(device)
    Pd = {}

    def f_prod(x, g, b):
        return P.batchnorm(x, g, b, mm_p, mv_p, True, relu, ggrad=g.gbuf, bgrad=b.gbuf, anchor=_anchor(x))

    def f_ref(x, g, b):
        Pd.update({'n/gamma': g, 'n/beta': b, 'n/moving_mean': mm0.double(), 'n/moving_variance': mv0.double()})
        upd = []
        y = O.batchnorm(x, Pd, 'n', True, upd)
        O.apply_bn_updates(Pd, upd)
        f_ref.pre = y
        return torch.relu(y) if relu else y

    check(f_prod, f_ref, [x, gamma, beta], device, rtol=5e-4, param_idx=(1, 2))
    _close(mm_p, Pd['n/moving_mean'], 'moving_mean')
    _close(mv_p, Pd['n/moving_variance'], 'moving_variance')


def test_batchnorm_infer(device):
    C = 64
    x = rnd(2, 8, 8, C, seed=9)
    g, b, mm, mv = rnd(C, seed=1) + 1, rnd(C, seed=2), rnd(C, seed=3), torch.rand(C) + 0.5
    y = P.batchnorm(x.to(device), g.to(device), b.to(device), mm.to(device), mv.to(device), False, True)
    Pd = {'n/gamma': g, 'n/beta': b, 'n/moving_mean': mm, 'n/moving_variance': mv}
    _close(y, torch.relu(O.batchnorm(x, Pd, 'n', False)), 'bn_infer')


def test_maxpool(device):
    x = rnd(2, 16, 12, 64, seed=10)
    x[0, :4, :4] = 0.0   # ties: gradient must go to the first maximum
    check(P.maxpool2, O.maxpool2, [x], device)


@pytest.mark.parametrize('C', [8, 5])
def test_softmax_round(C, device):
    x = rnd(2, 16, 16, C, seed=11) * 3

    def f_prod(x):
        p, s = P.softmax_round(x)
        return p, s

    def f_ref(x):
        p = torch.softmax(x, -1)
        return p, O.round_ste(p)

    check(f_prod, f_ref, [x], device)


@pytest.mark.parametrize('R,K,N,act', [(8, 28800, 32, 'leaky'), (8, 32, 8, None), (8, 1000, 100, 'tanh'),
                                       (8, 8, 2048, None), (8, 5000, 1, None), (16, 300, 50, None),
                                       (16, 46656, 1, None), (32, 46656, 1, None), (12, 4098, 1, None), (16, 5000, 3, 'leaky'),
                                       (32, 8, 8192, None), (5, 16, 1500, 'tanh')])       # few inputs, many outputs: the small-K data gradient
def test_dense(R, K, N, act, device):
    x, w, b = rnd(R, K, seed=12), rnd(K, N, seed=13, scale=K ** -0.5), rnd(N, seed=14, scale=0.1)

    def f_ref(x, w, b):
        y = O.dense(x, w, b)
        return O.leaky_relu(y, 0.3) if act == 'leaky' else (torch.tanh(y) if act == 'tanh' else y)

    check(lambda x, w, b: P.dense(x, w, b, act, 0.3, wgrad=w.gbuf, bgrad=b.gbuf, anchor=_anchor(x)), f_ref, [x, w, b], device,
          param_idx=(1, 2))


def test_film(device):
    x, g, b, r = rnd(2, 16, 16, 8, seed=15), rnd(2, 8, seed=16), rnd(2, 8, seed=17), rnd(2, 16, 16, 8, seed=18)
    check(lambda x, g, b, r: P.film(x, g, b, r, 0.3),
          lambda x, g, b, r: O.leaky_relu(O.film(x, g, b), 0.3) + r, [x, g, b, r], device)


def test_tps_warp(device):
    B, H, W, C = 2, 24, 20, 8
    vol = (torch.rand(B, H, W, C, generator=torch.Generator().manual_seed(19)) > 0.5).float() + rnd(B, H, W, C, seed=20) * 0.1
    theta = rnd(B, 25, 2, seed=21, scale=0.05)
    Mb = O.tps_basis(H, W)

    def f_prod(vol, theta):
        return P.tps_warp(vol, theta, Mb.float().to(vol.device))

    check(f_prod, lambda v, t: O.tps_warp(v, t), [vol, theta], device, rtol=1e-3)


def test_tps_identity(device):
    """KAT: theta = 0 is the identity warp (the Dense producing theta is zero-initialised, stn_spline.py:116)."""
    B, H, W, C = 1, 16, 16, 8
    vol = rnd(B, H, W, C, seed=22)
    out = P.tps_warp(vol.to(device), torch.zeros(B, 25, 2, device=device), O.tps_basis(H, W).float().to(device))
    _close(out, vol, 'identity', rtol=1e-5)


def test_maximum_slice_sampling_add(device):
    a = (rnd(2, 8, 8, 8, seed=23) > 0).float()
    b = (rnd(2, 8, 8, 8, seed=24) > 0).float()      # many exact ties
    check(P.maximum, lambda a, b: torch.where(a >= b, a, b), [a, b], device)
    x = rnd(2, 8, 8, 5, seed=25)
    check(lambda x: P.slice_channels(x, 0, 4), lambda x: x[..., 0:4], [x], device)
    mu, lv, eps = rnd(8, 8, seed=26), rnd(8, 8, seed=27) * 0.3, rnd(8, 8, seed=28)
    check(lambda m, l: P.sampling_kl(m, l, eps.to(m.device)), lambda m, l: (O.sampling(m, l, eps.double()), O.kl(m, l)),
          [mu, lv], device)
    check(P.add, lambda a, b: a + b, [a, x.new_ones(a.shape)], device)


@pytest.mark.parametrize('with_mod,act_alpha', [(True, 0.2), (True, -1.0), (False, -1.0)])
def test_instnorm_spade(with_mod, act_alpha, device):
    x = rnd(2, 8, 8, 16, seed=29) * 2 + 1
    ins = [x] + ([rnd(2, 8, 8, 16, seed=30) * 0.3, rnd(2, 8, 8, 16, seed=31) * 0.3] if with_mod else [])

    def f_ref(x, g=None, b=None):
        v = O.instance_norm(x)
        if g is not None:
            v = O.spade_cond(v, g, b)
        return O.leaky_relu(v, act_alpha) if act_alpha >= 0 else v

    check(lambda x, g=None, b=None: P.instnorm_spade(x, g, b, act_alpha), f_ref, ins, device, rtol=5e-4)


def _masks(B, H, W, nm, seed):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, nm + 1, (B, H, W), generator=g)
    return torch.nn.functional.one_hot(lab, nm + 1).float()


@pytest.mark.parametrize('lam', [0.01, 0.0])
def test_seg_loss(lam, device):
    B, H, W, nm = 2, 16, 16, 4
    t = _masks(B, H, W, nm, 32)
    pred = torch.softmax(rnd(B, H, W, nm + 1, seed=33), -1)
    pr = pred.clone().double().requires_grad_(True)
    ref = O.combined_dice_bce(t.double(), pr, nm) if lam else O.dice_loss(t.double(), pr, nm)
    (gref,) = torch.autograd.grad(ref * 10.0, pr)
    loss, dp = P.seg_loss(pred.to(device), t.to(device), nm, lam, 10.0)
    _close(loss, ref.reshape(1), 'loss', 1e-5)
    _close(dp, gref, 'dpred', 1e-4)


@pytest.mark.parametrize('mode', ['mae', 'mse', 'mean'])
def test_diff_loss(mode, device):
    p, t = rnd(2, 8, 8, 3, seed=34), rnd(2, 8, 8, 3, seed=35)
    pr = p.clone().double().requires_grad_(True)
    ref = {'mae': O.mae(t.double(), pr), 'mse': O.mse(t.double(), pr), 'mean': pr.mean()}[mode]
    (gref,) = torch.autograd.grad(ref * 0.5, pr)
    loss, dp = P.diff_loss(p.to(device), t.to(device), mode, 0.5)
    _close(loss, ref.reshape(1), 'loss', 1e-5)
    _close(dp, gref, 'grad', 1e-5)
    loss1, _ = P.diff_loss(p.to(device), 1.0, 'mse', 1.0)
    _close(loss1, O.mse(torch.ones_like(p), p).reshape(1), 'mse-const', 1e-5)


def test_spectral_reg(device):
    w = rnd(4, 4, 16, 32, seed=36, scale=0.1)
    u0 = torch.rand(256, 1, generator=torch.Generator().manual_seed(37)) * 2 - 1
    wr = w.clone().double().requires_grad_(True)
    ref = O.spectral_reg(wr, u0.double(), 10.0)
    (gref,) = torch.autograd.grad(ref, wr)
    loss, sgn = P.spectral_reg(w.to(device), u0.to(device), 10.0)
    _close(loss, ref.reshape(1), 'loss', 1e-4)
    _close(P.spectral_reg_grad(w.to(device), sgn), gref, 'grad', 1e-5)


def test_adam(device):
    n = 1003
    p, g = rnd(n, seed=38), rnd(n, seed=39)
    Pd = {'p': p.clone().double()}
    opt = O.KerasAdam(1e-3)
    pp, m, v = p.clone().to(device), torch.zeros(n, device=device), torch.zeros(n, device=device)
    import math
    for t in range(1, 4):
        gt = g * t
        opt.step(Pd, {'p': gt.double()})
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        P.adam_step(pp, gt.to(device), m, v, lr_t)
    _close(pp, Pd['p'], 'adam', 1e-6)


# ---- in-graph per-sample losses of the automated-pairing trainers ----------------------------------------------------
def _anat(B, H, W, C, seed):
    """one-hot-ish {0,1} anatomies like the rounded encoder output"""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, C + 1, (B, H, W), generator=g)
    return torch.nn.functional.one_hot(idx, C + 1)[..., :C].float()


def test_overlap_dice(device):
    B, H, W, C = 3, 16, 12, 8
    ref = _anat(B, H, W, C, 1) * 0.9 + 0.05
    others = [_anat(B, H, W, C, s) * 0.8 + 0.1 for s in (2, 3, 4)]
    check(lambda r, a, b, c: P.overlap_dice(r, [a, b, c]),
          lambda r, a, b, c: torch.cat([O.pair_dice(r, x) for x in (a, b, c)], dim=1), [ref] + others, device)


def test_row_mae(device):
    x, y = rnd(3, 16, 12, 1, seed=5), rnd(3, 16, 12, 1, seed=6)
    check(lambda a, b: P.row_mae(a, b), lambda a, b: O.mae_single_input(a, b).reshape(-1), [x, y], device,
          grad_mask=[False, True])


def test_seg_loss_per_sample(device):
    B, H, W = 3, 16, 12
    t = torch.cat([_anat(B, H, W, 4, 7), torch.zeros(B, H, W, 1)], -1)
    t[..., 4] = 1 - t[..., :4].sum(-1)
    t[0, :3] = t[0, :3] * 0.5 + 0.1                      # fractional labels (rotated masks) go through the softmax too
    p = torch.softmax(rnd(B, H, W, 5, seed=8), -1)
    check(lambda a, b: P.seg_loss_per_sample(a, b, 4), lambda a, b: O.combined_dice_bce_perbatch(a, b, 4), [t, p], device,
          grad_mask=[False, True])


def test_row_dot(device):
    w = torch.softmax(rnd(4, 3, seed=9), -1)
    ls = [rnd(4, 1, seed=10), rnd(4, seed=11), rnd(4, 1, seed=12)]
    check(lambda w_, a, b, c: P.row_dot(w_, [a, b, c]),
          lambda w_, a, b, c: w_[:, 0:1] * a + w_[:, 1:2] * b.reshape(-1, 1) + w_[:, 2:3] * c, [w] + ls, device)


@pytest.mark.parametrize('C1,C2,Cout,ups,relu', [(64, 0, 64, False, True), (1, 0, 64, False, True), (128, 0, 64, True, False),
                                                  (64, 64, 64, False, True), (8, 0, 64, False, True)])
def test_conv_bn_infer_fused(C1, C2, Cout, ups, relu, device):
    """`predict` of Conv2D -> BatchNormalization [-> ReLU] in one launch == the oracle's two layers with moving statistics"""
    B, H, W = 2, 16, 16
    x1 = rnd(B, H // 2 if ups else H, W // 2 if ups else W, C1, seed=1)
    x2 = rnd(B, H, W, C2, seed=2) if C2 else None
    w = rnd(3, 3, C1 + C2, Cout, seed=3, scale=(2.0 / (9 * (C1 + C2))) ** 0.5)
    cb, gamma, beta = rnd(Cout, seed=4, scale=0.1), torch.rand(Cout) + 0.5, rnd(Cout, seed=5, scale=0.2)
    mm, mv = rnd(Cout, seed=6, scale=0.3), torch.rand(Cout) + 0.3
    d = lambda t: t.to(device) if t is not None else None
    with torch.no_grad():
        y = P.conv2d_bn_infer(d(x1), d(w), d(cb), d(gamma), d(beta), d(mm), d(mv), relu=relu, x2=d(x2), upsample=ups)
    D = lambda t: t.double()
    xin = O.upsample2(D(x1)) if ups else D(x1)
    if x2 is not None:
        xin = torch.cat([xin, D(x2)], -1)
    Pd = {'bn/gamma': D(gamma), 'bn/beta': D(beta), 'bn/moving_mean': D(mm), 'bn/moving_variance': D(mv)}
    ref = O.batchnorm(O.conv2d(xin, D(w), D(cb)), Pd, 'bn', False)
    if relu:
        ref = torch.relu(ref)
    _close(y, ref, 'conv+bn(infer)')


def _bf16(t, dt=torch.bfloat16):
    return t.float().to(dt).to(torch.float64)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('B,H,W,Cin,C2,Cout,k,stride,padding,ups', [
    (2, 16, 16, 64, 0, 64, 3, 1, 'same', False), (2, 32, 32, 128, 0, 128, 3, 1, 'same', False),
    (2, 16, 16, 64, 64, 64, 3, 1, 'same', False), (2, 16, 16, 128, 0, 64, 3, 1, 'same', True),
    (2, 8, 8, 256, 0, 512, 3, 1, 'same', False), (2, 16, 16, 64, 0, 128, 4, 2, 'valid', False),
    (3, 24, 24, 64, 0, 8, 1, 1, 'same', False), (3, 17, 19, 64, 0, 5, 1, 1, 'same', False), (3, 17, 19, 8, 0, 1, 1, 1, 'same', False)])
def test_conv2d_bf16_precision(B, H, W, Cin, C2, Cout, k, stride, padding, ups, mode):
    """mmseg_set_conv_precision(1): forward and data gradient == the fp64 oracle on bf16-rounded operands (the products are
    then exact, only the fp32 accumulation differs); the same for the weight gradient."""
    dev = 'cuda'
    tdt = torch.bfloat16 if mode == 'bf16' else torch.float16
    _bf16 = lambda t: globals()['_bf16'](t, tdt)
    x1 = rnd(B, H // 2 if ups else H, W // 2 if ups else W, Cin, seed=1)
    x2 = rnd(B, H, W, C2, seed=2) if C2 else None
    w = rnd(k, k, Cin + C2, Cout, seed=3, scale=(2.0 / (k * k * (Cin + C2))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.1)
    D = lambda t: t.double()
    xr1 = D(x1).requires_grad_(True)
    xr2 = D(x2).requires_grad_(True) if C2 else None
    wr = D(w).requires_grad_(True)
    xin = O.upsample2(xr1) if ups else xr1
    if C2:
        xin = torch.cat([xin, xr2], -1)
    # forward on rounded operands (rounding has zero derivative, so the straight-through form keeps the graph)
    rnd_st = lambda t: t + (_bf16(t.detach()) - t.detach())
    yr = O.conv2d(rnd_st(xin), rnd_st(wr), D(b), stride=stride, padding=padding)
    cot = rnd(*yr.shape, seed=5)
    prev = P.set_conv_precision(mode)
    try:
        xp1 = x1.to(dev).requires_grad_(True)
        xp2 = x2.to(dev).requires_grad_(True) if C2 else None
        wp, bp = w.to(dev), b.to(dev)
        wg, bg = torch.zeros_like(wp), torch.zeros_like(bp)
        y = P.conv2d(xp1, wp, bp, stride=stride, padding=padding, x2=xp2, upsample=ups, wgrad=wg, bgrad=bg, anchor=_anchor(xp1))
        y.backward(cot.to(dev))
    finally:
        P.set_conv_precision(prev)
    _close(y, yr, 'bf16 forward')
    # data gradient: the kernel rounds the incoming gradient and the weights -- on the fast path (Cout % 32 == 0) and in the
    # 16-bit variant of the generic kernel (stride-1 launches with 4-channel gathers: Cout % 4 == 0); other shapes are fp32 only
    r_ = _bf16 if (Cout % 32 == 0 or (Cout % 4 == 0 and stride == 1)) else (lambda t: t.double())
    gx = torch.autograd.grad(O.conv2d(xin, r_(wr.detach()), None, stride=stride, padding=padding), [xr1] + ([xr2] if C2 else []),
                             r_(cot))
    _close(xp1.grad, gx[0], 'bf16 dgrad x1', 4e-4)
    if C2:
        _close(xp2.grad, gx[1], 'bf16 dgrad x2', 4e-4)
    # weight gradient (fast path: channel counts divisible by 4): input patches and incoming gradient rounded to bf16
    rw = _bf16 if ((Cin + C2) % 4 == 0 and Cout % 4 == 0) else (lambda t: t.double())      # other shapes: fp32 weight-gradient kernels
    gw = torch.autograd.grad(O.conv2d(rw(xin.detach()), wr, None, stride=stride, padding=padding), wr, rw(cot))[0]
    _close(wg, gw, 'bf16 wgrad', 4e-4)
    # and the rounding is really happening: the result differs from the unrounded oracle by more than fp32 noise
    y32 = O.conv2d(xin.detach(), D(w), D(b), stride=stride, padding=padding)
    if Cin >= 32 and Cout % 4 == 0:
        assert (y.detach().cpu().double() - y32).abs().max() > (1e-4 if mode == 'bf16' else 1e-5) * y32.abs().max()


@pytest.mark.parametrize('n', [4, 2])
def test_spectral_reg_multi(n, device):
    """the Spectral penalties of a discriminator's down-sample blocks in one batch of launches == one matrix at a time"""
    shapes = [(4, 4, 4, 64), (4, 4, 64, 128), (4, 4, 128, 256), (4, 4, 256, 512)][:n]
    ws_, u0s, refs, grefs = [], [], [], []
    for i, shp in enumerate(shapes):
        w = rnd(*shp, seed=40 + i, scale=0.05)
        u0 = torch.rand(shp[0] * shp[1] * shp[2], 1, generator=torch.Generator().manual_seed(50 + i)) * 2 - 1
        wr = w.clone().double().requires_grad_(True)
        ref = O.spectral_reg(wr, u0.double(), 10.0)
        refs.append(ref.reshape(1))
        grefs.append(torch.autograd.grad(ref, wr)[0])
        ws_.append(w.to(device))
        u0s.append(u0.to(device))
    loss, sgn = P.spectral_reg_multi(ws_, u0s, 10.0)
    _close(loss, torch.cat(refs), 'losses', 1e-4)
    grads = [torch.full_like(w, 0.5) for w in ws_]             # accumulates into existing gradients
    P.spectral_reg_grad_accumulate(ws_, sgn, grads)
    for g, gr in zip(grads, grefs):
        _close(g - 0.5, gr, 'grad', 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_weight_images_of_a_model_in_one_launch(mode):
    """mmseg_conv2d_wprep_batch (the forward / data-gradient images of all of a model's kernels after its optimiser step) == one
    mmseg_conv2d_wprep launch per image, bit for bit; odd channel counts and tap counts, both layouts, both element types"""
    from multimodal_segmentation_amd import _native as N
    dev = torch.device('cuda')
    shapes = [(3, 3, 64, 64, 0), (3, 3, 64, 64, 1), (4, 4, 128, 256, 0), (1, 1, 40, 36, 1), (5, 5, 20, 20, 0), (3, 3, 96, 8, 1), (2, 3, 33, 65, 0)]
    prev = P.set_conv_precision(mode)
    try:
        ws_, refs, outs, rows, blk = [], [], [], [], 0
        for i, (KH, KW, Cin, Cout, m) in enumerate(shapes):
            w = rnd(KH, KW, Cin, Cout, seed=70 + i).to(dev)
            ref, out = torch.zeros(w.numel(), device=dev), torch.zeros(w.numel(), device=dev)
            N.call('mmseg_conv2d_wprep', w, ref, KH, KW, Cin, Cout, m)
            rows.append([w.data_ptr(), out.data_ptr(), KH * KW, Cin, Cout, m, blk, 0])
            blk += KH * KW * ((Cin + 31) // 32) * ((Cout + 31) // 32)
            ws_.append(w); refs.append(ref); outs.append(out)
        rows.append([0, 0, 0, 0, 0, 0, blk, 0])
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        N.call('mmseg_conv2d_wprep_batch', table, len(shapes), blk)
        torch.cuda.synchronize()
        for shp, ref, out in zip(shapes, refs, outs):
            assert torch.equal(ref, out), shp
    finally:
        P.set_conv_precision(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('multi_stream', [False, True])
def test_training_iterations_with_batched_weight_images_are_bitwise_the_lazy_ones(multi_stream):
    """three DAFNet iterations with the per-model image refresh after each optimiser step == the same iterations with every image
    re-laid out by its own launch on first use (ops._wprep_batch_on off), bit for bit in every weight arena"""
    from tests import helpers as Hh
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor

    def run(batched):
        P._wprep_batch_on[0] = batched
        P.bump_weight_version()
        nn.set_default_device('cuda:0')
        np.random.seed(77)
        conf = Hh.make_conf(dafnet_config_chaos, 64, batch_size=4, multi_stream=multi_stream)
        model = DAFNet(conf)
        model.build()
        model.Enc_Modality._eps_rng = None
        ex = DAFNetExecutor(conf, model)
        np.random.seed(78)
        ex.init_train_data(slices_per_volume=3)
        losses = {n: [] for n in ex.get_loss_names()}
        for _ in range(3):
            ex.train_batch(losses)
        torch.cuda.synchronize()
        registered = sum(len(b['ents']) for b in P._wprep_batch.values())
        ms = model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]
        return [m.arena.detach().clone() for m in ms], registered

    try:
        a, na = run(True)
        b, _ = run(False)
    finally:
        P._wprep_batch_on[0] = True
    assert na > 20                                   # the images were registered (and refreshed in batches)
    assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))

