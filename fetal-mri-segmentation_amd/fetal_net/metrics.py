"""Loss / metric tokens with the reference names (reference fetal_net/metrics.py:7-100).

The builders look losses up by name (`getattr(fetal_net.metrics, config['loss'])`, reference fetal/train_fetal.py:31)
and compare them by identity (`loss_function != dice_coefficient_loss`, reference unet3d/unet.py:82), so each is a
module-level callable.  Inside a training step the Dice loss and the compiled metrics are computed on the device by
fmri_sigmoid_dice_fwd/bwd; called directly with arrays these functions evaluate the same formulas on the host (numpy,
float64) for evaluation scripts.
"""
from functools import partial

import numpy as np


def _f(a):
    return np.asarray(a, dtype=np.float64)


def dice_coefficient(y_true, y_pred, smooth=1.):
    yt, yp = _f(y_true).ravel(), _f(y_pred).ravel()
    return (2. * np.sum(yt * yp) + smooth) / (np.sum(yt) + np.sum(yp) + smooth)


def vod_coefficient(y_true, y_pred, binarize=True, smooth=1.):
    yt, yp = _f(y_true).ravel(), _f(y_pred).ravel()
    if binarize:
        yt, yp = (yt > 0.5).astype(np.float64), (yp > 0.5).astype(np.float64)
    inter = np.sum(yt * yp)
    return (inter + smooth) / (np.sum(yt) + np.sum(yp) - inter + smooth)


def dice_coefficient_loss(y_true, y_pred):
    return -dice_coefficient(y_true, y_pred)


def vod_coefficient_loss(y_true, y_pred):
    return -vod_coefficient(y_true, y_pred, binarize=False)


def double_dice_loss(y_true, y_pred, ratio=10.0):
    return -dice_coefficient(y_true, y_pred) + ratio * dice_coefficient(1 - _f(y_true), y_pred)


def weighted_dice_coefficient(y_true, y_pred, axis=(-3, -2, -1), smooth=0.00001):
    yt, yp = _f(y_true), _f(y_pred)
    return np.mean(2. * (np.sum(yt * yp, axis=axis) + smooth / 2) / (np.sum(yt, axis=axis) + np.sum(yp, axis=axis) + smooth))


def weighted_dice_coefficient_loss(y_true, y_pred):
    return -weighted_dice_coefficient(y_true, y_pred)


def label_wise_dice_coefficient(y_true, y_pred, label_index):
    return dice_coefficient(_f(y_true)[..., label_index], _f(y_pred)[..., label_index])


def get_label_dice_coefficient_function(label_index):
    f = partial(label_wise_dice_coefficient, label_index=label_index)
    f.__setattr__('__name__', 'label_{0}_dice_coef'.format(label_index))
    return f


def weighted_cross_entropy_loss(y_true, y_pred, weight_mask=None):
    yt = _f(y_true)
    yp = np.clip(_f(y_pred), 1e-7, 1 - 1e-7)
    xent = -(yt * np.log(yp) + (1 - yt) * np.log(1 - yp))
    if weight_mask is not None:
        xent = weight_mask * xent
    return np.mean(xent)


def binary_crossentropy(y_true, y_pred):
    yt = _f(y_true)
    yp = np.clip(_f(y_pred), 1e-7, 1 - 1e-7)
    return np.mean(-(yt * np.log(yp) + (1 - yt) * np.log(1 - yp)), axis=-1)


def dice_and_xent(y_true, y_pred, xent_weight=1.0, weight_mask=None):
    return dice_coef_loss(y_true, y_pred) + xent_weight * weighted_cross_entropy_loss(y_true, y_pred, weight_mask)


def _focal_loss(gamma=2., alpha=.5):
    def focal_loss_fixed(y_true, y_pred):
        yt, yp = _f(y_true), _f(y_pred)
        pt_1 = np.where(yt == 1, yp, np.ones_like(yp))
        pt_0 = np.where(yt == 0, yp, np.zeros_like(yp))
        return -np.sum(alpha * (1. - pt_1) ** gamma * np.log(pt_1)) - np.sum((1 - alpha) * pt_0 ** gamma * np.log(1. - pt_0))

    return focal_loss_fixed


class MaskInput(object):
    """placeholder for the model's second input (the distance mask) that `dice_and_xent_mask` closes over in the reference
    (isensee2017.py:85-88: loss_function = loss_function(mask_input))"""


def dice_and_xent_mask(weight_mask, xent_weight=1.0, dist_sigma=3):
    def _loss(y_true, y_pred):
        return dice_and_xent(y_true, y_pred, xent_weight=xent_weight, weight_mask=np.exp(-_f(weight_mask) / dist_sigma))

    # what the device path needs to know: Dice + xent_weight * mean(exp(-mask / dist_sigma) * cross-entropy), mask = 2nd model input
    _loss.mask_weighted = isinstance(weight_mask, MaskInput)
    _loss.xent_weight, _loss.dist_sigma = float(xent_weight), float(dist_sigma)
    _loss.__name__ = "dice_and_xent_mask"
    return _loss


dice_coef = dice_coefficient
dice_coef_loss = dice_coefficient_loss
binary_crossentropy_loss = binary_crossentropy
focal_loss = _focal_loss()

# losses differentiated on the device (fmri_sigmoid_loss_bwd); the rest build but raise at the first training step
DEVICE_LOSSES = (dice_coefficient_loss, binary_crossentropy_loss, dice_and_xent, focal_loss, vod_coefficient_loss, double_dice_loss,
                 weighted_dice_coefficient_loss)
