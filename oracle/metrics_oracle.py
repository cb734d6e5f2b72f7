"""numpy restatement of reference fetal_net/metrics.py:7-100 (test infrastructure).

PINNED: checked against tests/golden/metrics_golden.json, which holds the outputs of the reference's own
metrics.py (imported under a numpy keras.backend shim by tests/golden/make_fixtures.py) including the three
known-answer tests of reference test/test_metrics.py:10-38.
"""
import numpy as np


def dice_coefficient(y_true, y_pred, smooth=1.0):                       # metrics.py:11-15
    yt = np.asarray(y_true, np.float64).reshape(-1)
    yp = np.asarray(y_pred, np.float64).reshape(-1)
    inter = np.sum(yt * yp)
    return (2.0 * inter + smooth) / (np.sum(yt) + np.sum(yp) + smooth)


def dice_coefficient_loss(y_true, y_pred):                              # metrics.py:31-32
    return -dice_coefficient(y_true, y_pred)


def vod_coefficient(y_true, y_pred, binarize=True, smooth=1.0):         # metrics.py:18-28
    yt = np.asarray(y_true, np.float64).reshape(-1)
    yp = np.asarray(y_pred, np.float64).reshape(-1)
    if binarize:
        yt = (yt > 0.5).astype(np.float64)
        yp = (yp > 0.5).astype(np.float64)
    inter = np.sum(yt * yp)
    union = np.sum(yt) + np.sum(yp) - inter
    return (inter + smooth) / (union + smooth)


def vod_coefficient_loss(y_true, y_pred):                               # metrics.py:35-36
    return -vod_coefficient(y_true, y_pred, binarize=False)


def weighted_dice_coefficient(y_true, y_pred, axis=(-3, -2, -1), smooth=0.00001):   # metrics.py:39-51
    yt = np.asarray(y_true, np.float64)
    yp = np.asarray(y_pred, np.float64)
    return np.mean(2.0 * (np.sum(yt * yp, axis=axis) + smooth / 2) / (np.sum(yt, axis=axis) + np.sum(yp, axis=axis) + smooth))


def weighted_dice_coefficient_loss(y_true, y_pred):
    return -weighted_dice_coefficient(y_true, y_pred)


def double_dice_loss(y_true, y_pred, ratio=10.0):                       # metrics.py:7-8
    yt = np.asarray(y_true, np.float64)
    return -dice_coefficient(yt, y_pred) + ratio * dice_coefficient(1 - yt, y_pred)


def weighted_cross_entropy_loss(y_true, y_pred, weight_mask=None):      # metrics.py:73-77 (K.binary_crossentropy clips at 1e-7)
    yt = np.asarray(y_true, np.float64)
    yp = np.clip(np.asarray(y_pred, np.float64), 1e-7, 1 - 1e-7)
    xent = -(yt * np.log(yp) + (1 - yt) * np.log(1 - yp))
    if weight_mask is not None:
        xent = weight_mask * xent
    return np.mean(xent)


def dice_and_xent(y_true, y_pred, xent_weight=1.0, weight_mask=None):   # metrics.py:68-70
    return dice_coefficient_loss(y_true, y_pred) + xent_weight * weighted_cross_entropy_loss(y_true, y_pred, weight_mask)


def focal_loss(y_true, y_pred, gamma=2.0, alpha=0.5):                   # metrics.py:80-87
    yt = np.asarray(y_true, np.float64)
    yp = np.asarray(y_pred, np.float64)
    pt_1 = np.where(yt == 1, yp, np.ones_like(yp))
    pt_0 = np.where(yt == 0, yp, np.zeros_like(yp))
    return -np.sum(alpha * (1.0 - pt_1) ** gamma * np.log(pt_1)) - np.sum((1 - alpha) * pt_0 ** gamma * np.log(1.0 - pt_0))


def binary_accuracy(y_true, y_pred):                                    # Keras metric 'binary_accuracy' (unet.py:81)
    return float(np.mean(np.round(np.asarray(y_pred, np.float64)) == np.asarray(y_true, np.float64)))


def hard_dice(truth, pred):                                             # reference fetal/evaluate.py:13-17
    t = np.asarray(truth) > 0
    p = np.asarray(pred) > 0
    return 2.0 * np.sum(t & p) / (np.sum(t) + np.sum(p))
