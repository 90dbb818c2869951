"""hipGraph capture of a trainer step (`conf.hip_graphs`, `bench.py --graphs`; build-defined, the reference has no counterpart:
Keras/TF-1 runs its whole step inside one session call, which is what a captured graph restores here).

One `Trainer.fit` is forward + losses + backward + regulariser terms + Adam over several hundred to a thousand kernel launches that
the host queues one by one (ctypes call, argument marshalling, autograd bookkeeping).  For a fixed signature of the call (shapes of
the inputs / targets, scalar targets) the sequence of launches is always the same, so after two eager warm-up steps the third is
recorded into a graph (torch.cuda.CUDAGraph = hipGraph on ROCm; every kernel of the library is launched on torch's current stream,
which during the recording is the capturing stream) and every later step replays it with one host call.  What varies from step to
step lives in device memory the recorded kernels read:

  * the inputs and tensor targets of the call  -> static buffers the replay copies into first (the executors' fake pools, pure
    inference passes, are recorded the same way: `GraphedCall`);
  * values the step draws on the host (the noise of the sampling layer, `host_draw`) -> static buffers refilled from the same host
    generators in the same order, so a replayed run consumes exactly the random streams of an eager run;
  * Adam's bias-corrected step size lr_t        -> one device float (`mmseg_adam_p`), written before the replay.

Python-side effects of a step (optimiser iteration count, weight / BatchNorm-state version counters that invalidate cached weight
images) are repeated by the replay wrapper.  Results are bit-identical to the eager path (`tests/test_graph_capture.py`).
Not used under data parallelism (the RCCL all-reduces issued during backward are left out of captures)."""
import itertools
import weakref

import numpy as np
import torch

_serial = itertools.count(1)

WARMUP = 2          # eager steps before the recording (allocations, workspaces and caches reach their steady state)

_recording = None   # the FitGraph in discovery or capture mode, or None


class _Draw(object):
    __slots__ = ('fn', 'buf')

    def __init__(self, fn, buf):
        self.fn, self.buf = fn, buf


def host_draw(fn, device):
    """A float32 array the step draws on the host (fn() -> numpy array) as a device tensor.  Eager: upload of fn().  While a step is
    being recorded: the next static buffer (already holding this step's draw); the replay refills the buffers in call order."""
    st = _recording
    from . import nn
    if st is None:
        return nn.host_to_device(fn(), device)       # asynchronous (a pageable `.to(device)` stalls the host until the GPU has drained)
    if st.mode == 'discover':
        v = nn.host_to_device(fn(), device)
        st.draws.append(_Draw(fn, torch.empty_like(v)))
        return v
    d = st.draws[st.draw_i]
    st.draw_i += 1
    return d.buf


class GraphedCall(object):
    """A no-grad device function of tensors -> tuple of tensors (the executors' fake pools: inference passes of the generator's
    components on a fresh batch) recorded per input-shape signature after WARMUP eager calls.  The returned tensors are the
    graph's static outputs: valid until the next call of the same signature (the pools are sampled right away).  Host draws inside
    (the sampling layer's noise) go through `host_draw` as in a trainer step."""

    def __init__(self, fn):
        self.fn = fn
        self.states = {}

    def __call__(self, *xs):
        global _recording
        from . import ops
        from .parallel import dp
        if dp.enabled() or not all(isinstance(x, torch.Tensor) and x.is_cuda for x in xs):
            return self.fn(*xs)
        key = tuple((tuple(x.shape), x.dtype) for x in xs)
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = FitGraph(None)
        st.calls += 1
        if st.calls <= WARMUP:
            if st.calls == WARMUP:
                st.mode, st.draws = 'discover', []
                _recording = st
                try:
                    return self.fn(*xs)
                finally:
                    _recording = None
            return self.fn(*xs)
        if st.graph is None:
            st.s_in = [torch.empty_like(x) for x in xs]
        for b, x in zip(st.s_in, xs):
            b.copy_(x)
        for d in st.draws:
            FitGraph._fill(d.buf, d.fn())
        if st.graph is None:
            ops.bump_weight_version()                 # every cached weight image / BatchNorm fold is recomputed INSIDE the recording
            st.mode, st.draw_i = 'capture', 0
            _recording = st
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g), torch.no_grad():
                    st.outs = self.fn(*st.s_in)
            finally:
                _recording = None
            assert st.draw_i == len(st.draws)
            st.graph = g
        st.graph.replay()
        ops.bump_weight_version()                     # the images the replay wrote are not tagged with a host-side version
        return st.outs


def _sig(x):
    if x is None:
        return None
    if isinstance(x, (int, float)):
        return ('s', float(x))
    if isinstance(x, (list, tuple)):
        return tuple(_sig(v) for v in x)
    return ('t', tuple(x.shape))


def signature(inputs, targets):
    return (_sig(inputs), _sig(targets))


class FitGraph(object):
    """the recorded step of one trainer for one call signature"""

    def __init__(self, trainer):
        from . import ops
        self.t = trainer
        self.uid = next(_serial)       # names the scratch buffers the recording owns (ops._sid); never reused, unlike id()
        weakref.finalize(self, ops.release_graph_workspaces, self.uid)
        self.calls = 0
        self.mode = None
        self.draws, self.draw_i = [], 0
        self.graph = None
        self.launches_hint = None

    @staticmethod
    def _static_like(x, device):
        if isinstance(x, torch.Tensor):
            return torch.empty(x.shape, dtype=torch.float32, device=device)
        return torch.empty(np.shape(x), dtype=torch.float32, device=device)

    @staticmethod
    def _fill(buf, x):
        if isinstance(x, torch.Tensor):
            buf.copy_(x)
        else:       # staged through pinned memory, asynchronous on the stream (a pageable source would block the host per replay)
            from . import nn
            buf.copy_(nn.host_to_device(x, buf.device), non_blocking=True)

    def _is_tensorlike(self, x):
        return isinstance(x, (torch.Tensor, np.ndarray))

    def run(self, inputs, targets):
        global _recording
        t = self.t
        self.calls += 1
        if self.calls <= WARMUP:
            if self.calls == WARMUP:                 # the last eager step also lists the host draws of a step
                self.mode, self.draws = 'discover', []
                _recording = self
                try:
                    return t._fit_eager(inputs, targets, {})
                finally:
                    _recording = None
            return t._fit_eager(inputs, targets, {})
        dev = t.device
        if self.graph is None:
            self.s_in = [self._static_like(x, dev) for x in inputs]
            self.s_tg = [self._static_like(x, dev) if self._is_tensorlike(x) else x for x in targets]
        for b, x in zip(self.s_in, inputs):
            self._fill(b, x)
        for b, x in zip(self.s_tg, targets):
            if self._is_tensorlike(x):
                self._fill(b, x)
        for d in self.draws:                          # this step's host draws, in the order the step asks for them
            FitGraph._fill(d.buf, d.fn())
        lr_dev = t.optimizer.begin_device_step(dev)   # iteration count + 1, lr_t of this step into the device scalar
        if self.graph is None:
            from . import ops as _ops
            _ops.bump_weight_version()               # every cached weight image is recomputed INSIDE the recording
            self.mode, self.draw_i = 'capture', 0
            _recording = self
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    hist = t._fit_eager(self.s_in, self.s_tg, {}, lr_dev=lr_dev)
                    names = list(hist._dev.keys())
                    vals, self.total_terms = [], None
                    for n in names:
                        v = hist._dev[n]
                        if hasattr(v, 'terms'):      # the lazy total: (weight, device scalar) pairs
                            self.total_terms = [(w, len(vals) + i) for i, (w, _) in enumerate(v.terms)]
                            vals.extend(x.reshape(1) for _, x in v.terms)
                        else:
                            vals.append(v.reshape(1))
                    self.names = names
                    self.loss_vec = torch.cat(vals)
            finally:
                _recording = None
            assert self.draw_i == len(self.draws), 'the recorded step drew %d host values, the eager step %d' % (self.draw_i, len(self.draws))
            self.graph = g
            self.outs = t.last_outputs
        self.graph.replay()
        t.last_outputs = self.outs
        from . import nn, ops
        ops.bump_weight_version()                     # what optimizer.step does on the host side of an eager step
        vec = self.loss_vec.clone()                   # the static scalars are overwritten by the next replay
        hist = nn.History()
        i = 0
        for n in self.names:
            if n == 'loss' and self.total_terms is not None:
                hist.record(n, t._total([(w, vec[j]) for w, j in self.total_terms]))
                i += len(self.total_terms)
            else:
                hist.record(n, vec[i])
                i += 1
        return hist
