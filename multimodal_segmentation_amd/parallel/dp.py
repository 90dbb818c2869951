"""Per-slice data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The path shards by independent slice pairs: every rank holds a full replica of the weights and processes its own
batch; the only exchanges are
  (i)  an all-reduce (mean) of each trainer's gradient arenas -- flat, 16-byte aligned fp32 buffers
       (nn.Model.grad_arena), one collective per arena SEGMENT (layer-aligned, contiguous ranges of >= SEGMENT_FLOATS
       floats, nn.Model.grad_segments: the 119 MB arena of the shared UNet up path is six collectives in the order its
       layers finish in the backward pass, every other component one or two) -- OVERLAPPED WITH THE BACKWARD PASS: every kernel launch that accumulates into an arena is counted in the forward pass
       (GradTracker.register, from nn.Param.g) and un-counted when its backward has been queued (GradTracker.done, from
       the autograd Functions of ops.py); when a model's count returns to zero its arena is final and its all-reduce is
       issued at once (async: RCCL runs it on its own stream behind the kernels queued so far) while the backward pass
       of the remaining models (the decoder / segmentor finish first, the encoders last) keeps the compute stream busy;
  (ii) an all-reduce (sum) of the 2x8 batch-global class sums of the swapped-argument BCE (costs.py:77-79 of the
       reference computes its class weights over the whole batch), so the loss equals the single-device loss on
       the global batch;
  (iii) a broadcast of the initial weights / non-trainable state from rank 0.
BatchNorm uses per-rank (local) batch statistics ("ghost batch norm" with the per-GPU batch), see DESIGN.md.
With world size 1 every function here is a no-op.
"""
import torch
import torch.distributed as dist

_state = {'enabled': False}

# A gradient arena is all-reduced in layer-aligned segments of at least this many floats (16 MB): large enough for a ring over xGMI's
# point-to-point links to run at its bandwidth, small enough that the collectives of a big arena (shared UNet up path: 119 MB, whose
# last layer -- the bottleneck -- finishes just before the first encoder's down path) start while its other layers are still in the
# backward pass instead of all at once behind the last of them.
SEGMENT_FLOATS = 4 * 1024 * 1024


def enable(flag=True, force=False):
    """`force`: keep the collectives on even with a single rank (used to exercise the RCCL path on a one-GPU box)"""
    _state['enabled'] = bool(flag) and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def enabled():
    return _state['enabled']


def world_size():
    return dist.get_world_size() if enabled() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def is_main():
    """rank 0 (or no process group): the only rank that writes checkpoints, csv files and logs"""
    return rank() == 0


def init_from_env(backend=None):
    """One process per GPU, launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment): bind this process to its GPU, join the RCCL ("nccl") group and switch data parallelism on.  Returns
    (rank, local_rank, world).  Without WORLD_SIZE > 1 in the environment this is a no-op returning (0, LOCAL_RANK or 0, 1).
    Must run before the models are built (they are allocated on the default device)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world <= 1:
        return 0, local_rank, 1
    use_gpu = torch.cuda.is_available()
    backend = backend or ('nccl' if use_gpu else 'gloo')
    if use_gpu:
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kw = {'device_id': torch.device('cuda', local_rank)} if (use_gpu and backend == 'nccl') else {}
        dist.init_process_group(backend, **kw)
    if _state.get('host_group') is None:
        # a CPU-side group for waits of unbounded length (host_barrier): created collectively, here, by every rank -- also when the
        # default group is gloo already (CPU runs): its 30-minute default timeout is the kind this group exists to avoid
        import datetime
        _state['host_group'] = dist.new_group(backend='gloo', timeout=datetime.timedelta(hours=48))
    if use_gpu:
        from .. import nn
        nn.set_default_device('cuda:%d' % local_rank)
    enable(True)
    return dist.get_rank(), local_rank, dist.get_world_size()


def sync_model(model):
    """broadcast every component of a DAFNet / MMSDNet wrapper from rank 0 (weights, BatchNorm moving statistics, the
    spectral regularisers' u0)"""
    if not enabled():
        return
    ms = list(model._generator_models())
    for name in ('D_Mask', 'D_Image1', 'D_Image2', 'Balancer'):
        m = getattr(model, name, None)
        if m is not None and m not in ms:
            ms.append(m)
    broadcast_models(ms)


def barrier():
    if enabled():
        dist.barrier()


def host_barrier():
    """wait for every rank WITHOUT a GPU collective: rank 0 evaluates the test split alone after training (experiment.py) while the
    other ranks wait here -- a RCCL barrier would hit the process group's 10-minute watchdog on a real-size test pass.  Uses the
    gloo side group created by init_from_env (48 h timeout, whatever the default backend); process groups set up by hand fall back
    to the default group's barrier."""
    if not enabled():
        return
    g = _state.get('host_group')
    if g is not None:
        dist.barrier(group=g)
    else:
        dist.barrier()


def average_state(models):
    """all-reduce(mean) of the non-trainable state (BatchNorm moving statistics; the spectral u0 are equal already) so that
    the replicas -- which normalise with per-rank batch statistics during an epoch -- validate, early-stop and checkpoint
    identically at the epoch boundary"""
    if not enabled():
        return
    from .. import ops
    ws = float(dist.get_world_size())
    for m in models:
        t = m.state_arena
        if t is not None and t.numel() > 0:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ops.axpby(t, t, 1.0 / ws, 0.0, out=t)
    ops.bump_weight_version()


def set_sync_bn(flag):
    """conf.sync_bn (build-defined, default False): training-mode BatchNorm statistics over the GLOBAL batch of all ranks instead
    of per-rank "ghost" statistics -- a data-parallel step then equals the single-device step on the concatenated batch"""
    _state['sync_bn'] = bool(flag)


def sync_bn():
    return enabled() and _state.get('sync_bn', False)


def all_gather_rows(local):
    """[n] tensor of every rank -> [world, n] in rank order (identical on every rank)"""
    out = torch.empty((dist.get_world_size(),) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous()) if hasattr(dist, 'all_gather_into_tensor') and local.is_cuda else \
        dist.all_gather(list(out.unbind(0)), local.contiguous())
    return out


def all_reduce_sum(t):
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def class_sum_hook():
    if not enabled():
        return None

    def hook(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return hook


_tracker = [None]


def current_tracker():
    return _tracker[0]


class GradTracker(object):
    """Issues the all-reduce of a segment of a model's gradient arena as soon as the last kernel accumulating into it has been queued.
    Everything here happens on the host thread that queues the backward pass, in the order the autograd engine visits the graph --
    a function of the graph alone, so every rank issues the same collectives in the same order whatever its timing (`order` is kept
    for the tests that check exactly that: a mismatch would be a RCCL deadlock on hardware)."""

    def __init__(self, models, defer=()):
        self.models = list(models)
        self.pending = {}
        for m in self.models:
            for sg in range(len(m.grad_segments)):
                self.pending[(m.uid, sg)] = 0
        self.by_uid = {m.uid: m for m in self.models}
        self.used = set()
        self.defer = set(m.uid for m in defer)      # arenas that still receive gradients after backward (regularisers)
        self.fired = set()
        self.works = []
        self.order = []

    def register(self, owner, seg=0):
        k = (owner.uid, seg)
        if k in self.pending:
            self.pending[k] += 1
            self.used.add(k)

    def done(self, owner, seg=0):
        k = (owner.uid, seg)
        if k not in self.pending:
            return
        self.pending[k] -= 1
        if self.pending[k] == 0 and owner.uid not in self.defer and k not in self.fired:
            self._fire(owner, seg)

    def _fire(self, m, seg):
        self.fired.add((m.uid, seg))
        a, b = m.grad_segments[seg]
        view = m.grad_arena[a:b]
        self.order.append((m.name, seg))
        self.works.append((view, dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)))

    def finish(self):
        """reduce whatever has not been reduced yet (in model and segment order: identical on every rank), wait, average"""
        _state['last_overlapped'] = len(self.works)     # collectives issued during the backward pass
        _state['total_overlapped'] = _state.get('total_overlapped', 0) + len(self.works)
        for m in self.models:
            for sg in range(len(m.grad_segments)):
                if (m.uid, sg) in self.used and (m.uid, sg) not in self.fired:
                    self._fire(m, sg)
        _state['last_collectives'] = len(self.works)
        _state['last_order'] = list(self.order)
        import hashlib
        _state['order_digest'] = hashlib.sha1((_state.get('order_digest', '') + repr(self.order)).encode()).hexdigest()
        _state['total_collectives'] = _state.get('total_collectives', 0) + len(self.works)
        _state['total_steps'] = _state.get('total_steps', 0) + 1
        from .. import ops
        ws = float(dist.get_world_size())
        for view, w in self.works:
            w.wait()
            ops.axpby(view, view, 1.0 / ws, 0.0, out=view)


def n_segments(models):
    """collectives one trainer step over these models issues (every segment of every arena once)"""
    return sum(len(m.grad_segments) for m in models)


def begin(models, defer=()):
    """start tracking one trainer step; returns None when data parallelism is off"""
    if not enabled():
        return None
    _tracker[0] = GradTracker(models, defer)
    return _tracker[0]


def finish(tracker):
    if tracker is None:
        return
    _tracker[0] = None
    tracker.finish()


def counters(reset=False):
    """gradient collectives since the last reset: {'steps': trainer steps, 'collectives': arena all-reduces, 'overlapped': those issued
    while the backward pass was still being queued, 'order_digest': a hash chain over the (arena, segment) sequence of every step, equal on
    all ranks iff they issued the same collectives in the same order} -- the evidence a first multi-GPU run prints beside its throughput"""
    out = {'steps': _state.get('total_steps', 0), 'collectives': _state.get('total_collectives', 0),
           'overlapped': _state.get('total_overlapped', 0), 'order_digest': _state.get('order_digest', '')}
    if reset:
        _state['total_steps'] = _state['total_collectives'] = _state['total_overlapped'] = 0
        _state['order_digest'] = ''
    return out


def replica_checksums(models):
    """[2 * len(models)] float64: (sum, sum of squares) of every model's weight arena AND non-trainable state on this rank"""
    vals = []
    for m in models:
        for t in (m.arena, m.state_arena):
            d = t.detach().double() if (t is not None and t.numel() > 0) else torch.zeros(1, dtype=torch.float64)
            vals += [d.sum().reshape(1), (d * d).sum().reshape(1)]
    return torch.cat(vals)


def replicas_identical(models, check_state=False):
    """-> (ok, local checksum list): every rank holds bit-identical weight arenas (all-reduce MAX == all-reduce MIN of the per-arena
    checksums).  BatchNorm moving statistics are per-rank between epoch boundaries (ghost statistics), so the non-trainable state is
    only included on request (after dp.average_state).  Verification plumbing of bench.py / the tests, not part of the step."""
    cs = replica_checksums(models)
    if not check_state:
        keep = torch.tensor([i for i in range(cs.numel()) if (i // 2) % 2 == 0])
        cs = cs[keep]
    if not enabled():
        return True, cs.tolist()
    dev = models[0].arena.device if dist.get_backend() == 'nccl' else torch.device('cpu')
    hi, lo = cs.to(dev).clone(), cs.to(dev).clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo)), cs.tolist()


def broadcast_models(models, src=0):
    if not enabled():
        return
    for m in models:
        for t in (m.arena, m.state_arena):
            if t is not None and t.numel() > 0:          # components without non-trainable state have an empty arena
                dist.broadcast(t, src=src)
    from .. import ops
    ops.bump_weight_version()
