"""conf.hip_graphs (graphs.py): a trainer step replayed from a recorded hipGraph is bit-identical to the eager step -- weights, Adam
state, BatchNorm moving statistics and every loss of five consecutive steps (two eager warm-up steps, the recording, two replays),
for a discriminator trainer and for the full DAFNet generator trainer (whose sampling layer draws its noise on the host)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(hip_graphs, decoder='film', **extra):
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    conf = Hh.make_conf(dafnet_config_chaos, 64, hip_graphs=hip_graphs, decoder_type=decoder, **extra)
    model = DAFNet(conf)
    model.build()
    return model


def _state(models):
    return [w.copy() for m in models for w in m.get_weights()]


@pytest.mark.parametrize('extra', [{}, {'compute_dtype': 'bf16', 'act_storage': 'half'},
                                   {'compute_dtype': 'bf16', 'act_storage': 'half', 'decoder': 'spade'}],
                         ids=['fp32', 'bf16-act16', 'spade-bf16-act16'])
def test_discriminator_and_generator_steps_replayed_from_graphs_are_bitwise_the_eager_steps(extra):
    from multimodal_segmentation_amd import ops as P
    from tests import helpers as Hh
    B, H, steps = 2, 64, 5
    rng = np.random.RandomState(3)
    data = [Hh.make_step_data(B, H, H, seed=40 + i) for i in range(steps)]
    fake = [rng.rand(B, H, H, 4).astype(np.float32) for _ in range(steps)]
    runs = {}
    ref_w = None
    for mode in (False, True):
        model = _build(mode, **extra)
        ms = model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]
        if ref_w is None:
            ref_w = [m.get_weights() for m in ms]
        else:
            for m, w in zip(ms, ref_w):
                m.set_weights(w)
        model.Enc_Modality._eps_rng = None            # both runs start the private noise stream from its seed
        assert model.supervised_trainer.use_graph == mode and model.D_Mask_trainer.use_graph == mode
        losses = []
        B1 = np.ones((B, 1), np.float32)
        for i in range(steps):
            d = data[i]
            h = model.D_Mask_trainer.fit([d['m1'][..., :4].copy(), fake[i]], [1.0, 0.0])
            losses.append(('dm%d' % i, h.history['loss'][0]))
            tg = [d['m1'], d['m2'], d['m1'], d['m2']] + [B1] * 4 + [d['x1'], d['x2'], d['x1'], d['x2']] + [B1] * 4 + \
                 [np.zeros(B, np.float32)] * 2 + [d['z1'], d['z2']]
            h = model.supervised_trainer.fit([d['x1'], d['x2'], d['z1'], d['z2']], tg)
            for k in h.history.keys():
                losses.append(('g%d/%s' % (i, k), h.history[k][0]))
        if mode:
            g = model.supervised_trainer._graphs
            assert len(g) == 1 and list(g.values())[0].graph is not None, 'the generator step was not recorded'
            assert len(list(g.values())[0].draws) >= 2, 'the host draws of the sampling layer were not listed'
        runs[mode] = (losses, _state(ms), model.supervised_trainer.optimizer.iterations)
    P.set_activation_storage(False)
    P.set_conv_precision('fp32')
    (l0, w0, it0), (l1, w1, it1) = runs[False], runs[True]
    assert it0 == it1 == steps
    for (k0, v0), (k1, v1) in zip(l0, l1):
        assert k0 == k1 and v0 == v1, 'loss %s: eager %r, graph %r' % (k0, v0, v1)
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('switch', ['hip_graphs', 'multi_stream', 'hip_graphs+multi_stream'])
def test_executor_iterations_with_graphs_are_bitwise_the_eager_iterations(switch):
    """the whole DAFNetExecutor.train_batch with conf.hip_graphs (trainer steps AND the fake pools replayed from graphs) or with
    conf.multi_stream (the two discriminator phases on concurrent HIP streams): five iterations end in bit-identical weights and
    per-iteration losses"""
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    runs = {}
    ref_w = None
    for mode in (False, True):
        np.random.seed(123)
        conf = Hh.make_conf(dafnet_config_chaos, 64, batch_size=4, **{k: mode for k in switch.split('+')})
        model = DAFNet(conf)
        model.build()
        ms = model._generator_models() + [model.D_Mask, model.D_Image1, model.D_Image2]
        if ref_w is None:
            ref_w = [m.get_weights() for m in ms]
        else:
            for m, w in zip(ms, ref_w):
                m.set_weights(w)
        model.Enc_Modality._eps_rng = None
        ex = DAFNetExecutor(conf, model)
        np.random.seed(321)
        ex.init_train_data(slices_per_volume=3)
        losses = {n: [] for n in ex.get_loss_names()}
        for _ in range(5):
            ex.train_batch(losses)
        if mode and 'multi_stream' in switch:
            assert getattr(ex, '_streams', None) is not None, 'the concurrent-stream path did not run'
        if mode and 'hip_graphs' in switch:
            from multimodal_segmentation_amd import graphs
            assert isinstance(ex.mask_pools, graphs.GraphedCall) and any(st.graph is not None for st in ex.mask_pools.states.values())
            assert any(st.graph is not None for st in ex.image_pools.states.values())
        runs[mode] = ({k: [float(v.item()) if hasattr(v, 'item') else float(v) for v in vs] for k, vs in losses.items()}, _state(ms))
    (l0, w0), (l1, w1) = runs[False], runs[True]
    assert l0 == l1, 'per-iteration losses differ between the eager and the graph-replayed executor'
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('nmod,l_mix,switch', [(2, 1.0, 'multi_stream'), (3, 0.5, 'multi_stream'), (2, 1.0, 'hip_graphs+multi_stream')])
def test_mmsdnet_iterations_on_concurrent_streams_are_bitwise_the_single_stream_iterations(nmod, l_mix, switch):
    """MMSDNetExecutor.train_batch with conf.multi_stream (the last Z-regressor step beside the mask-discriminator phase): four
    iterations end in bit-identical weights"""
    from multimodal_segmentation_amd import nn
    from multimodal_segmentation_amd.configuration import mmsdnet_config_chaos, mmsdnet3_config_chaos
    from multimodal_segmentation_amd.models.mmsdnet import MMSDNet
    from multimodal_segmentation_amd.model_executors.mmsdnet_executor import MMSDNetExecutor
    from tests import helpers as Hh
    nn.set_default_device('cuda:0')
    runs, ref_w = {}, None
    for mode in (False, True):
        np.random.seed(11)
        conf = Hh.make_conf(mmsdnet_config_chaos if nmod == 2 else mmsdnet3_config_chaos, 64, batch_size=2, l_mix=l_mix,
                            **{k: mode for k in switch.split('+')})
        model = MMSDNet(conf)
        model.build()
        ms = model._all_component_models() if hasattr(model, '_all_component_models') else []
        assert ms
        if ref_w is None:
            ref_w = [m.get_weights() for m in ms]
        else:
            for m, w in zip(ms, ref_w):
                m.set_weights(w)
        model.Enc_Modality._eps_rng = None
        ex = MMSDNetExecutor(conf, model)
        np.random.seed(12)
        ex.init_train_data(slices_per_volume=2)
        losses = {n: [] for n in ex.get_loss_names()}
        for _ in range(5 if 'hip_graphs' in switch else 4):
            ex.train_batch(losses)
        if mode:
            assert getattr(ex, '_streams', None) is not None
            if 'hip_graphs' in switch:
                assert any(st.graph is not None for st in model.supervised_trainer._graphs.values()), 'the generator step was not recorded'
        runs[mode] = _state(ms)
    for a, b in zip(runs[False], runs[True]):
        assert np.array_equal(a, b)
