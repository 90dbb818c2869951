"""DAFNet / SPADE decoder on CHAOS (reference configuration/dafnet_spade_config_chaos.py): the FiLM config with
folder 'dafnet_spade_chaos' and decoder_type 'spade'."""
from . import dafnet_config_chaos as _base


def get():
    p = _base.get()
    p['folder'] = 'dafnet_spade_chaos'
    p['decoder_type'] = 'spade'
    return p
