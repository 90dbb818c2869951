"""MMSDNet on CHAOS (keys and values of reference configuration/mmsdnet_config_chaos.py:3-53): reconstruction weight 10,
a 4-filter mask discriminator, no image discriminators."""
from . import _chaos


def get():
    return _chaos.assemble('mmsdnet_chaos', 'mmsdnet.MMSDNet', 'mmsdnet_executor.MMSDNetExecutor', d_mask_filters=4,
                           w_rec_X=10)
