"""The data-parallel machinery on the REAL backend: torch.distributed "nccl" (= RCCL) with ONE rank on the GPU box.

With a single rank every all-reduce is the identity, so a step with data parallelism forced on must reproduce the plain
step BITWISE -- which it only does if the collectives (issued on RCCL's stream from GradTracker in the middle of the
backward pass) are ordered correctly against the ctypes kernel launches on the compute stream: an all-reduce that ran
before its arena's last accumulation, or an Adam update that ran before the reduction landed, would change the weights."""
import socket

import numpy as np
import pytest
import torch

from tests import helpers as Hh

H = 64


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_steps(force_dp):
    import torch.distributed as dist
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.parallel import dp
    nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H)
    model = DAFNet(conf)
    model.build()
    dp.enable(force_dp, force=force_dp)
    assert dp.enabled() == force_dp
    gens = model._generator_models()
    if force_dp:
        dp.sync_model(model)
    B = 2
    d = Hh.make_step_data(B, H, H, seed=5)
    ones = np.ones((B, 1), np.float32)
    out = {}
    for it in range(2):
        model.D_Mask_trainer.fit([d['dm_m1'], d['dm_m2']], [1.0, 0.0])
        if force_dp:
            assert dp._state['last_overlapped'] == 0 and dp._state['last_collectives'] == 1
        h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']],
                                         [d['m1'], d['m2'], d['m1'], d['m2']] + [ones] * 4 +
                                         [d['x1'], d['x2'], d['x1'], d['x2']] + [ones] * 4 +
                                         [np.zeros(B, np.float32)] * 2 + [d['z1'], d['z2']],
                                         eps=[d['eps1'], d['eps2']])
        if force_dp:     # every generator arena reduced while the backward pass was still being queued
            assert dp._state['last_overlapped'] == dp.n_segments(gens) == dp._state['last_collectives'], dp._state
        out['loss%d' % it] = h.history['loss'][0]
    torch.cuda.synchronize()
    out['gen'] = [m.arena.detach().cpu().numpy().copy() for m in gens]
    out['dm'] = model.D_Mask.arena.detach().cpu().numpy().copy()
    dp.enable(False)
    return out


@pytest.mark.gpu
def test_rccl_single_rank_step_is_bitwise_the_plain_step():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    ref = _run_steps(False)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        assert dist.get_backend() == 'nccl'
        got = _run_steps(True)
    finally:
        dist.destroy_process_group()
    assert got['loss0'] == ref['loss0'] and got['loss1'] == ref['loss1'], (got['loss0'], ref['loss0'], got['loss1'], ref['loss1'])
    assert np.array_equal(got['dm'], ref['dm']), 'D_Mask weights differ after two RCCL-reduced steps'
    for a, b in zip(got['gen'], ref['gen']):
        assert np.array_equal(a, b), 'generator weights differ after two RCCL-reduced steps'


@pytest.mark.gpu
def test_sync_batchnorm_kernels_single_rank_equal_plain_batchnorm():
    """conf.sync_bn over RCCL with one rank: the split kernels (local statistics -> all-gather -> combine; local sums -> all-reduce
    -> finish -> apply) must reproduce the fused BatchNorm kernels on the same batch (the exact multi-rank equivalence with the
    global-batch step is tests/test_dp_gloo.py's, on the CPU stand-in)."""
    import torch.distributed as dist
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.models.trainer import Trainer, OutputSpec
    from multimodal_segmentation_amd.parallel import dp
    torch.cuda.set_device(0)
    nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, H)
    model = DAFNet(conf)
    model.build()
    seg = model.Segmentor
    w0 = seg.get_weights()
    s_in = (np.random.RandomState(12).rand(4, H, H, 8) > 0.7).astype(np.float32)
    d = Hh.make_step_data(4, H, H, seed=6)

    def step():
        seg.set_weights(w0)
        tr = Trainer('seg_only', lambda ins, training=True: [seg(ins[0], training=training)], [OutputSpec('Segmentor', 'dice_bce', 10.0)],
                     [seg], nn.Adam(1e-4), num_masks=4)
        h = tr.fit([s_in], [d['m1']])
        torch.cuda.synchronize()
        return h.history['loss'][0], seg.grad_arena.detach().cpu().numpy().copy(), seg.state_arena.detach().cpu().numpy().copy()
    ref = step()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        dp.enable(True, force=True)
        dp.set_sync_bn(True)
        assert dp.sync_bn()
        got = step()
    finally:
        dp.set_sync_bn(False)
        dp.enable(False)
        dist.destroy_process_group()
    assert abs(got[0] - ref[0]) <= 1e-5 * max(1.0, abs(ref[0]))
    assert np.abs(got[1] - ref[1]).max() <= 2e-5 * max(1.0, np.abs(ref[1]).max()), np.abs(got[1] - ref[1]).max()
    assert np.abs(got[2] - ref[2]).max() <= 1e-5, np.abs(got[2] - ref[2]).max()


def _run_executor(force_dp, multi_stream):
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from multimodal_segmentation_amd.parallel import dp
    nn.set_default_device('cuda:0')
    np.random.seed(77)
    conf = Hh.make_conf(dafnet_config_chaos, H, batch_size=4, multi_stream=multi_stream)
    model = DAFNet(conf)
    model.build()
    dp.enable(force_dp, force=force_dp)
    if force_dp:
        dp.sync_model(model)
    model.Enc_Modality._eps_rng = None
    ex = DAFNetExecutor(conf, model)
    np.random.seed(78)
    ex.init_train_data(slices_per_volume=3)
    losses = {n: [] for n in ex.get_loss_names()}
    for _ in range(3):
        ex.train_batch(losses)
    torch.cuda.synchronize()
    ms = model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]
    out = [m.arena.detach().cpu().numpy().copy() for m in ms]
    dp.enable(False)
    return out


@pytest.mark.gpu
def test_rccl_single_rank_iterations_with_concurrent_discriminator_streams_are_bitwise_the_plain_iterations():
    """conf.multi_stream under data parallelism: the discriminator phases run on side streams, their gradient all-reduces are queued
    from those streams (RCCL work objects waited on by the stream that issued them) -- three whole iterations equal the plain
    single-stream, no-DP iterations bit for bit"""
    import torch.distributed as dist
    torch.cuda.set_device(0)
    ref = _run_executor(False, False)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        got = _run_executor(True, True)
    finally:
        dist.destroy_process_group()
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
