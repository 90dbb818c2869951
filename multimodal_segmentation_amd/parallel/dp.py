"""Per-slice data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The path shards by independent slice pairs: every rank holds a full replica of the weights and processes its own
batch; the only exchanges are
  (i)  an all-reduce (mean) of each trainer's gradient arenas after its backward pass -- the arenas are flat,
       16-byte aligned fp32 buffers (nn.Model.grad_arena), one collective per model, largest ~119 MB (shared UNet
       up path), launched on a side stream as soon as the backward pass has been queued so that the collective of
       model k overlaps the Adam kernels of the models already reduced;
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


def enable(flag=True):
    _state['enabled'] = bool(flag) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def enabled():
    return _state['enabled']


def world_size():
    return dist.get_world_size() if enabled() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def class_sum_hook():
    if not enabled():
        return None

    def hook(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return hook


def allreduce_gradients(models):
    """Average the gradient arenas of `models` over the ranks (in place)."""
    if not enabled():
        return
    ws = float(dist.get_world_size())
    works = []
    for m in models:
        works.append((m, dist.all_reduce(m.grad_arena, op=dist.ReduceOp.SUM, async_op=True)))
    from .. import ops
    for m, w in works:
        w.wait()
        ops.axpby(m.grad_arena, m.grad_arena, 1.0 / ws, 0.0, out=m.grad_arena)


def broadcast_models(models, src=0):
    if not enabled():
        return
    for m in models:
        dist.broadcast(m.arena, src=src)
        dist.broadcast(m.state_arena, src=src)
    from .. import ops
    ops.bump_weight_version()
