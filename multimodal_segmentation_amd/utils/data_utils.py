"""Host-side numpy helpers (reference utils/data_utils.py): pool sampling used inside the training step (125-129) and the
intensity / geometry helpers of the data containers (8-118).  Pinned by tests/golden/reference_helpers.npz."""
import numpy as np


def sample_indices(n, nb_samples, seed=-1):
    """The index draw of reference `sample`: np.random.choice(len(data), size=nb_samples, replace=False)."""
    if seed > -1:
        np.random.seed(seed)
    return np.random.choice(n, size=nb_samples, replace=False)


def sample(data, nb_samples, seed=-1):
    idx = sample_indices(len(data), nb_samples, seed)
    return np.array([data[i] for i in idx])


def rescale(array, min_value=-1, max_value=1):
    """Affine map of the whole array onto [min_value, max_value]; a constant array becomes min_value (data_utils.py:8-21)"""
    lo, hi = array.min(), array.max()
    if hi == lo:
        return array * 0 + min_value
    out = (max_value - min_value) * (array - float(lo)) / (hi - lo) + min_value
    assert out.max() == max_value and out.min() == min_value, '%d, %d' % (out.max(), out.min())
    return out


def normalise(image):
    """(x - median) / (inter-quartile range + 1e-12)  (data_utils.py:23-36)"""
    q25, q50, q75 = np.percentile(image, [25, 50, 75])
    out = (image - q50) / ((q75 - q25) + 1e-12)
    assert not np.any(np.isnan(out)), 'NaN values in normalised array'
    return out


def _axis_slice(ndim, axis, lo, hi):
    s = [slice(None)] * ndim
    s[axis] = slice(lo, hi)
    return tuple(s)


def _crop(image, dim, nb_pixels, mode):
    """data_utils.py:83-101.  'equal' removes ceil(diff/2) pixels from BOTH ends, so an odd difference leaves
    nb_pixels-1 pixels and crop_same then pads one back."""
    n = image.shape[dim]
    diff = n - nb_pixels
    if mode == 'equal':
        lo = int(np.ceil(diff / 2))
        hi = n - lo
    elif mode == 'right':
        lo, hi = 0, nb_pixels
    elif mode == 'left':
        lo, hi = diff, n
    else:
        raise ValueError('Unexpected mode: %s. Expected to be one of [equal, left, right].' % mode)
    if dim not in (1, 2):
        return None
    return image[_axis_slice(image.ndim, dim, lo, hi)]


def _pad(image, dim, nb_pixels, mode='edge'):
    """data_utils.py:104-122: floor(diff/2) before, the rest after; 'constant' pads with the array minimum."""
    if dim not in (1, 2):
        return None
    diff = nb_pixels - image.shape[dim]
    before = int(diff / 2)
    width = [(0, 0)] * image.ndim
    width[dim] = (before, int(diff - before))
    if mode == 'edge':
        return np.pad(image, width, 'edge')
    if mode == 'constant':
        return np.pad(image, width, 'constant', constant_values=np.min(image))
    raise Exception('Invalid pad mode: ' + mode)


def _fit(a, dim, n, mode, pad_mode):
    if a.shape[dim] > n:
        a = _crop(a, dim, n, mode)
    if a.shape[dim] < n:
        a = _pad(a, dim, n, pad_mode)
    return a


def crop_same(image_list, mask_list, size=(None, None), mode='equal', pad_mode='edge'):
    """Crop / pad every (image, mask) pair to a common spatial size (data_utils.py:39-80); a `None` size defaults to
    the smallest mask extent."""
    n1 = np.min([m.shape[1] for m in mask_list]) if size[0] is None else size[0]
    n2 = np.min([m.shape[2] for m in mask_list]) if size[1] is None else size[1]
    imgs, msks = [], []
    for im, m in zip(image_list, mask_list):
        for dim, n in ((1, n1), (2, n2)):
            m = _fit(m, dim, n, mode, pad_mode)
            im = _fit(im, dim, n, mode, pad_mode)
        imgs.append(im)
        msks.append(m)
    return imgs, msks
