"""Data-parallel path on CPU: world_size 2 (and full iterations at 4 and 8) over gloo with the TEST-ONLY kernel stand-in.

  * the discriminator trainers contain no BatchNorm, so 2 ranks x batch 1 must reproduce the single-process step on
    the concatenated batch 2 (gradient all-reduce mean == global-batch gradient);
  * the generator trainers use per-rank BatchNorm statistics by design, so there the test checks that the
    replicas stay bit-identical after a step and that the batch-global BCE class sums were exchanged.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

H = 48


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from tests import cpu_backend as cb, helpers as Hh
    cb.install()
    from multimodal_segmentation_amd import nn
    nn.set_default_device('cpu')
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    conf = Hh.make_conf(dafnet_config_chaos, H)
    model = DAFNet(conf)
    model.build()
    return conf, model, Hh


def _tiny_seg_trainer():
    """BN-free segmentation head (1x1 conv 8 -> 5 + softmax) under the DAFNet segmentation loss (Dice + 0.01 * swapped BCE, weight
    10): per-sample arithmetic only, so 2 ranks x batch 1 must give the single-process gradient of the batch of 2."""
    from multimodal_segmentation_amd import nn, ops
    from multimodal_segmentation_amd.models.trainer import Trainer, OutputSpec

    class Tiny(nn.Model):
        def __init__(self):
            super(Tiny, self).__init__('TinySeg')
            nn.conv_params(self, 'c', 1, 8, 5)
            self.finalize(np.random.RandomState(3))

        def forward(self, x, training=False):
            return ops.softmax(nn.conv(self, 'c', x))
    m = Tiny()
    tr = Trainer('tiny', lambda ins, training=True: [m(ins[0])], [OutputSpec('Segmentor', 'dice_bce', 10.0)], [m],
                 nn.Adam(1e-4), num_masks=4)
    return m, tr


def _segmentor_trainer(model):
    """the product's Segmentor (conv - BatchNorm - ReLU twice + softmax head) as a trainer of its own under Dice + 0.01 * BCE"""
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.models.trainer import Trainer, OutputSpec
    seg = model.Segmentor
    return Trainer('seg_only', lambda ins, training=True: [seg(ins[0], training=training)], [OutputSpec('Segmentor', 'dice_bce', 10.0)],
                   [seg], nn.Adam(1e-4), num_masks=4)


def _seg_input(B):
    return (np.random.RandomState(12).rand(B, H, H, 8) > 0.7).astype(np.float32)


def _tiny_input(B):
    return np.random.RandomState(11).standard_normal((B, H, H, 8)).astype(np.float32)


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException as exc:          # surface the failure instead of letting the parent wait for its timeout
        import traceback
        q.put((rank, 'error', traceback.format_exc(), repr(exc)))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    conf, model, Hh = _build()
    from multimodal_segmentation_amd.parallel import dp
    dp.enable(True)
    assert dp.world_size() == world
    gens = model._generator_models()
    dp.broadcast_models(gens + [model.D_Mask, model.D_Image1, model.D_Image2])
    d = Hh.make_step_data(2, H, H, seed=5)                       # the GLOBAL batch; this rank takes sample `rank`
    sl = slice(rank, rank + 1)
    # ---- discriminator step: must equal the single-process step on the global batch --------------------------
    h = model.D_Mask_trainer.fit([d['dm_m1'][sl], d['dm_m2'][sl]], [1.0, 0.0])
    dm_after = model.D_Mask.arena.clone()
    assert dp._state['last_overlapped'] == 0 and dp._state['last_collectives'] == 1     # regularised arena: reduced at the end
    # ---- synchronised BatchNorm: the BatchNorm-ed Segmentor on 2 ranks x batch 1 == the single-process step on the batch of 2 ----
    dp.set_sync_bn(True)
    assert dp.sync_bn()
    _segmentor_trainer(model).fit([_seg_input(2)[sl]], [d['m1'][sl]])
    seg_grad = model.Segmentor.grad_arena.clone().numpy()
    seg_state = model.Segmentor.state_arena.clone().numpy()
    dp.set_sync_bn(False)
    # ---- generator step: replicas stay in sync, class sums are global -----------------------------------------
    ones = np.ones((1, 1), np.float32)
    model.supervised_trainer.fit([d['x1'][sl], d['x2'][sl], d['z1'][sl], d['z2'][sl]],
                                 [d['m1'][sl], d['m2'][sl], d['m1'][sl], d['m2'][sl]] + [ones] * 4 +
                                 [d['x1'][sl], d['x2'][sl], d['x1'][sl], d['x2'][sl]] + [ones] * 4 +
                                 [np.zeros(1, np.float32)] * 2 + [d['z1'][sl], d['z2'][sl]],
                                 eps=[d['eps1'][sl], d['eps2'][sl]])
    # every generator arena was all-reduced while the backward pass was still being queued (overlap), none at the end
    assert dp._state['last_overlapped'] == dp.n_segments(gens) == dp._state['last_collectives'], dp._state
    # ---- combined Dice + swapped-argument BCE on a BN-free head: the averaged gradient must be the global-batch gradient ----
    tiny, ttr = _tiny_seg_trainer()
    dp.broadcast_models([tiny])
    s_in = _tiny_input(2)
    ttr.fit([s_in[sl]], [d['m1'][sl]])
    tiny_grad = tiny.grad_arena.clone().numpy()
    sig = torch.cat([m.arena.double().sum().reshape(1) for m in gens])
    gathered = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(gathered, sig)
    # ---- a FULL DAFNet iteration through the executor (generator fit, both pools, four discriminator fits) on per-rank data ----
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    conf.batch_size = 1
    ex = DAFNetExecutor(conf, model)
    ex.init_train_data(device_resident=False, slices_per_volume=1)
    all_models = gens + [model.D_Mask, model.D_Image1, model.D_Image2]
    ok0, _ = dp.replicas_identical(all_models)
    dp.counters(reset=True)
    losses = {n: [] for n in ex.get_loss_names()}
    ex.train_batch(losses)
    cnt = dp.counters()
    ok1, cs = dp.replicas_identical(all_models)
    # different batches per rank (rank-offset data seed), yet identical replicas afterwards; 5 trainer steps, one collective per arena
    full = dict(ok0=ok0, ok1=ok1, steps=cnt['steps'], collectives=cnt['collectives'], overlapped=cnt['overlapped'],
                finite=all(np.isfinite(float(v)) for k in losses for v in losses[k]), n_gens=dp.n_segments(gens), checksum=cs[:4])
    q.put((rank, dm_after.numpy(), [g.numpy() for g in gathered], float(h.history['loss'][0]), tiny_grad, seg_grad, seg_state, full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_dp_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(2)]
    for r in res:
        assert r[1] != 'error' if isinstance(r[1], str) else True, r[2]
    res = sorted(res, key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the full executor iteration: replicas identical before and after, 5 trainer steps (generator + 2 D_Mask + D_Image1/2),
    # one collective per generator arena + one per discriminator fit, the generator's all overlapped with its backward pass
    for r in res:
        f = r[7]
        assert f['ok0'] and f['ok1'] and f['finite'], f
        assert f['steps'] == 5 and f['collectives'] == f['n_gens'] + 4 and f['overlapped'] == f['n_gens'], f
    assert res[0][7]['checksum'] == res[1][7]['checksum']
    # replicas identical after both steps
    assert np.array_equal(res[0][1], res[1][1]), 'D_Mask replicas diverged'
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b), 'generator replicas diverged'
    # single-process reference on the global batch of 2
    try:
        _single_process_reference(res)
    finally:
        from tests import cpu_backend as cb
        cb.uninstall()


def _single_process_reference(res):
    conf, model, Hh = _build()
    d = Hh.make_step_data(2, H, H, seed=5)
    model.D_Mask_trainer.fit([d['dm_m1'], d['dm_m2']], [1.0, 0.0])
    ref = model.D_Mask.arena.numpy()
    # Dice + swapped-argument BCE head: mean of the per-rank gradients == gradient of the loss on the global batch
    assert np.array_equal(res[0][4], res[1][4])
    tiny, ttr = _tiny_seg_trainer()
    ttr.fit([_tiny_input(2)], [d['m1']])
    gref = tiny.grad_arena.numpy()
    # synchronised BatchNorm: gradients AND moving statistics of the DP step equal the global-batch step's
    assert np.array_equal(res[0][5], res[1][5]) and np.array_equal(res[0][6], res[1][6]), 'SyncBN replicas diverged'
    _segmentor_trainer(model).fit([_seg_input(2)], [d['m1']])
    sg, ss = model.Segmentor.grad_arena.numpy(), model.Segmentor.state_arena.numpy()
    assert np.abs(res[0][5] - sg).max() < 2e-5 * max(1.0, np.abs(sg).max()), \
        'SyncBN gradient differs from the global-batch gradient: %g vs scale %g' % (np.abs(res[0][5] - sg).max(), np.abs(sg).max())
    assert np.abs(res[0][6] - ss).max() < 1e-5, 'SyncBN moving statistics differ: %g' % np.abs(res[0][6] - ss).max()
    assert np.abs(res[0][4] - gref).max() < 1e-5 * max(1.0, np.abs(gref).max()), \
        'DP segmentation-loss gradient differs from the global-batch gradient: %g vs scale %g' % (
            np.abs(res[0][4] - gref).max(), np.abs(gref).max())
    # after one Adam step |delta| = lr wherever the gradient is not ~0: compare the step direction
    assert np.abs(res[0][1] - ref).max() < 2.5e-4 * 1e-0 * 0 + 2e-6, 'DP D step differs from the global-batch step'


def _worker4(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(2 if world <= 4 else 1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        conf, model, Hh = _build()
        from multimodal_segmentation_amd.parallel import dp
        from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
        dp.enable(True)
        gens = model._generator_models()
        all_models = gens + [model.D_Mask, model.D_Image1, model.D_Image2]
        dp.broadcast_models(all_models)
        conf.batch_size = 1
        ex = DAFNetExecutor(conf, model)
        ex.init_train_data(device_resident=False, slices_per_volume=1)
        ok0, _ = dp.replicas_identical(all_models)
        dp.counters(reset=True)
        # uneven host timing: the last rank is slow to issue every collective and rank 1 stalls before every trainer step.  The order of
        # the collectives is decided by the graph walk on the host thread, never by who is ready first (a mismatch = RCCL deadlock)
        import time
        gen_order = []
        fire0, finish0 = dp.GradTracker._fire, dp.GradTracker.finish
        if rank == world - 1:
            def slow_fire(self, m, seg):
                time.sleep(0.01)
                return fire0(self, m, seg)
            dp.GradTracker._fire = slow_fire

        def finish_logged(self):
            if rank == 1:
                time.sleep(0.05)
            out = finish0(self)
            if len(self.order) > 4:
                gen_order.append(list(self.order))
            return out
        dp.GradTracker.finish = finish_logged
        losses = {n: [] for n in ex.get_loss_names()}
        for _ in range(2):
            ex.train_batch(losses)
        cnt = dp.counters()
        ok1, cs = dp.replicas_identical(all_models)
        first = float(losses[ex.get_loss_names()[0]][0])
        q.put((rank, dict(ok0=ok0, ok1=ok1, steps=cnt['steps'], collectives=cnt['collectives'], overlapped=cnt['overlapped'],
                          finite=all(np.isfinite(float(v)) for k in losses for v in losses[k]), n_gens=dp.n_segments(gens), checksum=cs[:4], first=first, digest=cnt['order_digest'],
                          order=dp._state['last_order'], gen_order=gen_order)))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as exc:
        import traceback
        q.put((rank, 'error', traceback.format_exc(), repr(exc)))
        raise


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [4, 8])
def test_dp_gloo_full_iterations_keep_the_replicas_identical(world):
    """Four and eight ranks (the driver scales to 2, 4 and 8): two full DAFNet iterations through the executor on per-rank batches -- rank-offset
    data and noise seeds, one collective per arena and trainer step averaged over all ranks -- leave bit-identical replicas everywhere;
    the ranks did see different data (their first losses differ)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == 'error'), r[2]
    res = sorted(res, key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, f in res:
        assert f['ok0'] and f['ok1'] and f['finite'], f
        assert f['steps'] == 10 and f['collectives'] == 2 * (f['n_gens'] + 4) and f['overlapped'] == 2 * f['n_gens'], f
    # every rank issued the same collectives in the same order (hash chain over all 10 trainer steps), slow ranks included
    assert len(set(f['digest'] for _, f in res)) == 1 and res[0][1]['digest'], [f['digest'] for _, f in res]
    go = res[0][1]['gen_order'][0]
    assert all(f['gen_order'] == res[0][1]['gen_order'] for _, f in res)
    # the shared up path (119 MB) goes out in several layer-ordered pieces, its last layers (conv_anatomy, u0 ...) first and the
    # bottleneck last; the second encoder's down path is complete -- and on the wire -- before the shared path's last piece
    shared = [sg for name, sg in go if name == 'Enc_Anatomy_shared']
    assert len(shared) >= 4 and shared == sorted(shared, reverse=True), go
    names = [name for name, _ in go]
    last_shared = max(i for i, n in enumerate(names) if n == 'Enc_Anatomy_shared')
    assert any(n.startswith('Enc_Anatomy_') and n != 'Enc_Anatomy_shared' for n in names[:last_shared]), go
    for _, f in res:
        assert f['checksum'] == res[0][1]['checksum']
    assert len({round(f['first'], 6) for _, f in res}) > 1, 'every rank reported the same first loss: the ranks saw the same batch'
