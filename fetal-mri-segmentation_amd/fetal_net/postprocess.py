"""Binary clean-up of a probability volume (same entry points and defaults as reference fetal_net/postprocess.py:7-19): gaussian
smoothing, threshold, hole filling, largest connected component.  Host code on scipy.ndimage, as in the reference: it runs once per
volume on the result of the device inference and is not on the hot path."""
import numpy as np
from scipy import ndimage


def get_main_connected_component(data):
    """boolean mask of the largest 6-connected component of `data` (all False when there is none)"""
    components, count = ndimage.label(data)
    if count == 0:
        return np.zeros(components.shape, dtype=bool)          # the reference raises on the argmax of an empty list here
    voxels_per_component = np.bincount(components.ravel(), minlength=count + 1)[1:]
    return components == 1 + int(np.argmax(voxels_per_component))


def postprocess_prediction(pred, gaussian_std=1, threshold=0.5, fill_holes=True, connected_component=True):
    mask = ndimage.gaussian_filter(pred, gaussian_std) > threshold
    steps = []
    if fill_holes:
        steps.append(ndimage.binary_fill_holes)
    if connected_component:
        steps.append(get_main_connected_component)
    for step in steps:
        mask = step(mask)
    return mask
