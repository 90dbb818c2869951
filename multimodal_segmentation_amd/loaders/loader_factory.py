"""Loader registry (reference loaders/loader_factory.py).  Only the shape/label metadata of the CHAOS loader is needed
by the hot path (configuration/*.py read input_shape and num_masks from it); reading DICOM volumes is out of scope
(SURVEY section 2 row 26).  'synthetic' generates the seeded synthetic slices of SURVEY 8(d)."""


class ChaosLoader(object):
    """Metadata of reference loaders/chaos.py:26-33."""

    def __init__(self):
        self.input_shape = (192, 192, 1)
        self.num_masks = 4
        self.modalities = ['t1', 't2']
        self.name = 'chaos'


def init_loader(dataset):
    if dataset in ('chaos', 'synthetic'):
        return ChaosLoader()
    return None
