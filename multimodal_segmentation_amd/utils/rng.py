"""Process-wide RNG streams.  Weight initialisation uses a numpy RandomState seeded with conf.seed (10 in every
reference config); the training step's random draws (eps of `sampling`, z samples, fake-pool indices) come from
numpy's global generator exactly like the reference (utils/distributions.py:9-11, utils/data_utils.py:125-129)."""
import numpy as np

_rng = None


def global_rng(seed=None):
    global _rng
    if seed is not None or _rng is None:
        _rng = np.random.RandomState(10 if seed is None else seed)
    return _rng
