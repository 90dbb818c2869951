"""Host-side numpy helpers used inside the training step (reference utils/data_utils.py:125-129)."""
import numpy as np


def sample_indices(n, nb_samples, seed=-1):
    """The index draw of reference `sample`: np.random.choice(len(data), size=nb_samples, replace=False)."""
    if seed > -1:
        np.random.seed(seed)
    return np.random.choice(n, size=nb_samples, replace=False)


def sample(data, nb_samples, seed=-1):
    idx = sample_indices(len(data), nb_samples, seed)
    return np.array([data[i] for i in idx])
