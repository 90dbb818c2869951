"""DAFNet with the FiLM decoder on CHAOS (keys and values of reference configuration/dafnet_config_chaos.py:3-59)."""
from . import _chaos


def get():
    return _chaos.assemble('dafnet_chaos', 'dafnet.DAFNet', 'dafnet_executor.DAFNetExecutor', d_mask_filters=64,
                           d_image_filters=64, randomise=False, automatedpairing=False)
