"""DAFNet with the SPADE decoder on CHAOS (reference configuration/dafnet_spade_config_chaos.py)."""
from . import _chaos


def get():
    return _chaos.assemble('dafnet_spade_chaos', 'dafnet.DAFNet', 'dafnet_executor.DAFNetExecutor', d_mask_filters=64,
                           d_image_filters=64, decoder_type='spade', randomise=False, automatedpairing=False)
