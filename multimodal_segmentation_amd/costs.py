"""Loss vocabulary of the reference's costs.py as used by the trainers.  The device implementations are
ops.seg_loss / ops.diff_loss (csrc/loss.hip); this module keeps the reference's names for the trainers' loss tables
and the numpy Dice metric used by validation and testing (costs.py:31-41)."""
import numpy as np

lambda_bce = 0.01   # costs.py:10


def make_combined_dice_bce(num_classes):
    """-> loss-kind tag understood by models.trainer.Trainer (costs.py:129-136; NOTE the swapped-argument BCE)."""
    return 'dice_bce'


def make_dice_loss_fnc(restrict_chn=1):
    return 'dice'   # costs.py:59-67


ypred = 'ypred'     # costs.py:194-195


def dice(y_true, y_pred, binarise=False, smooth=1e-12):
    """numpy metric, reference costs.py:31-41"""
    y_pred = y_pred[..., 0:y_true.shape[-1]]
    if binarise:
        y_pred = np.round(y_pred)
    y_int = y_true * y_pred
    return np.mean((2 * np.sum(y_int, axis=(1, 2, 3)) + smooth)
                   / (np.sum(y_true, axis=(1, 2, 3)) + np.sum(y_pred, axis=(1, 2, 3)) + smooth))
