"""MMSDNet with THREE modalities (BASELINE.json config #5) -- a build-defined extension: the reference hard-wires two
modalities (models/mmsdnet.py:105,120-129,160-161); see models/mmsdnet.py for the all-ordered-pairs graph.  Every other key
and value is mmsdnet_config_chaos's (reference configuration/mmsdnet_config_chaos.py:3-53)."""
from . import _chaos


def get():
    return _chaos.assemble('mmsdnet3_chaos', 'mmsdnet.MMSDNet', 'mmsdnet_executor.MMSDNetExecutor', d_mask_filters=4,
                           w_rec_X=10, modality=['t1', 't2', 't3'])
