"""Shared assembly of the CHAOS experiment configurations.

The reference keeps one literal dict per configuration module (configuration/dafnet_config_chaos.py:3-59,
mmsdnet_config_chaos.py:3-53); the config *contract* is the set of keys and values of `get()`, so here the
three configurations are expressed as differences from one table."""
from ..loaders import loader_factory

# key -> value shared by every CHAOS configuration (values of reference configuration/*.py)
COMMON = dict(seed=10, epochs=500, batch_size=6, split=0, dataset_name='chaos', test_dataset='chaos', image_downsample=1,
              modality=['t1', 't2'], l_mix=1, decoder_type='film', num_z=8, lr=0.0001,
              w_sup_M=10, w_adv_M=1, w_rec_X=1, w_adv_X=1, w_rec_Z=1, w_kl=0.1)
ANATOMY_ENCODER = dict(normalise='batch', downsample=4, filters=64, out_channels=8, rounding=True)


def discriminator(name, filters, shape):
    return dict(filters=filters, lr=0.0001, name=name, input_shape=shape)


def assemble(folder, model, executor, d_mask_filters, d_image_filters=None, **overrides):
    loader = loader_factory.init_loader('chaos')
    p = dict(COMMON, folder=folder, model=model, executor=executor)
    p.update(overrides)
    r = p['image_downsample']
    h, w, c = loader.input_shape
    shape = (int(h / r), int(w / r), c)
    p['input_shape'] = shape
    p['num_masks'] = loader.num_masks
    enc = dict(ANATOMY_ENCODER, input_shape=shape)
    enc['output_shape'] = shape[:-1] + (enc['out_channels'],)
    p['anatomy_encoder'] = enc
    p['d_mask_params'] = discriminator('D_Mask', d_mask_filters, shape[:-1] + (loader.num_masks,))
    if d_image_filters is not None:
        p['d_image_params'] = discriminator('D_Image', d_image_filters, shape)
    return p
