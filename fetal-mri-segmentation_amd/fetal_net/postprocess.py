"""Binary clean-up of a probability volume (reference fetal_net/postprocess.py:7-19): gaussian smoothing, threshold, hole filling,
largest connected component.  Host code on scipy.ndimage, as in the reference: it runs once per volume on the result of the
device inference and is not on the hot path."""
import numpy as np
from scipy import ndimage


def get_main_connected_component(data):
    labeled, n = ndimage.label(data)
    if n == 0:
        return labeled == 1            # nothing segmented (the reference raises on argmax of an empty list here)
    sizes = ndimage.sum(np.ones_like(labeled), labeled, index=np.arange(1, n + 1))
    return labeled == (int(np.argmax(sizes)) + 1)


def postprocess_prediction(pred, gaussian_std=1, threshold=0.5, fill_holes=True, connected_component=True):
    pred = ndimage.gaussian_filter(pred, gaussian_std) > threshold
    if fill_holes:
        pred = ndimage.binary_fill_holes(pred)
    if connected_component:
        pred = get_main_connected_component(pred)
    return pred
