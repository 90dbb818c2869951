"""DAFNet / FiLM decoder on CHAOS (reference configuration/dafnet_config_chaos.py:3-59): same keys, same values."""
from ..loaders import loader_factory as chaos

params = {
    'seed': 10,
    'folder': 'dafnet_chaos',
    'epochs': 500,
    'batch_size': 6,
    'split': 0,
    'dataset_name': 'chaos',
    'test_dataset': 'chaos',
    'input_shape': chaos.ChaosLoader().input_shape,
    'image_downsample': 1,                            # downsample image size: used for testing
    'modality': ['t1', 't2'],                         # list of [source, target] modalities
    'model': 'dafnet.DAFNet',                         # model to load
    'executor': 'dafnet_executor.DAFNetExecutor',     # model trainer
    'l_mix': 1,                                       # amount of supervision for target modality
    'decoder_type': 'film',                           # decoder type - can be film or spade
    'num_z': 8,                                       # dimensions of the modality factor
    'w_sup_M': 10,
    'w_adv_M': 1,
    'w_rec_X': 1,
    'w_adv_X': 1,
    'w_rec_Z': 1,
    'w_kl': 0.1,
    'lr': 0.0001,
    'randomise': False,
    'automatedpairing': False,
}

# discriminator configs
d_mask_params = {'filters': 64, 'lr': 0.0001, 'name': 'D_Mask'}
d_image_params = {'filters': 64, 'lr': 0.0001, 'name': 'D_Image'}

anatomy_encoder_params = {
    'normalise': 'batch',    # normalisation layer - can be batch or instance
    'downsample': 4,         # number of downsample layers of UNet encoder
    'filters': 64,           # number of filters in the first convolutional layer
    'out_channels': 8,       # number of output channels - dimensions of the anatomy factor
    'rounding': True
}


def get():
    import copy
    p, dm, di, ae = (copy.deepcopy(d) for d in (params, d_mask_params, d_image_params, anatomy_encoder_params))
    shp = p['input_shape']
    ratio = p['image_downsample']
    shp = (int(shp[0] / ratio), int(shp[1] / ratio), shp[2])

    p['input_shape'] = shp
    p['num_masks'] = chaos.ChaosLoader().num_masks

    dm['input_shape'] = (shp[:-1]) + (chaos.ChaosLoader().num_masks,)
    di['input_shape'] = shp

    ae['input_shape'] = shp
    ae['output_shape'] = (shp[:-1]) + (ae['out_channels'],)

    p.update({'anatomy_encoder': ae, 'd_mask_params': dm, 'd_image_params': di})
    return p
