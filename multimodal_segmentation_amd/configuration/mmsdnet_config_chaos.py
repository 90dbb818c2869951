"""MMSDNet on CHAOS (reference configuration/mmsdnet_config_chaos.py:3-53): same keys, same values."""
from ..loaders import loader_factory as chaos

params = {
    'seed': 10,
    'folder': 'mmsdnet_chaos',
    'epochs': 500,
    'batch_size': 6,
    'split': 0,
    'dataset_name': 'chaos',
    'test_dataset': 'chaos',
    'input_shape': chaos.ChaosLoader().input_shape,
    'image_downsample': 1,
    'modality': ['t1', 't2'],
    'model': 'mmsdnet.MMSDNet',
    'executor': 'mmsdnet_executor.MMSDNetExecutor',
    'l_mix': 1,
    'decoder_type': 'film',
    'num_z': 8,
    'w_sup_M': 10,
    'w_adv_M': 1,
    'w_rec_X': 10,
    'w_adv_X': 1,
    'w_rec_Z': 1,
    'w_kl': 0.1,
    'lr': 0.0001,
}

d_mask_params = {'filters': 4, 'lr': 0.0001, 'name': 'D_Mask'}

anatomy_encoder_params = {
    'normalise': 'batch',
    'downsample': 4,
    'filters': 64,
    'out_channels': 8,
    'rounding': True
}


def get():
    import copy
    p, dm, ae = (copy.deepcopy(d) for d in (params, d_mask_params, anatomy_encoder_params))
    shp = p['input_shape']
    ratio = p['image_downsample']
    shp = (int(shp[0] / ratio), int(shp[1] / ratio), shp[2])
    p['input_shape'] = shp
    p['num_masks'] = chaos.ChaosLoader().num_masks
    dm['input_shape'] = (shp[:-1]) + (chaos.ChaosLoader().num_masks,)
    ae['input_shape'] = shp
    ae['output_shape'] = (shp[:-1]) + (ae['out_channels'],)
    p.update({'anatomy_encoder': ae, 'd_mask_params': dm})
    return p
