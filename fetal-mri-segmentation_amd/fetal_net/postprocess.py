"""Binary clean-up of a probability volume (same entry points and defaults as reference fetal_net/postprocess.py:7-19): gaussian
smoothing, threshold, hole filling, largest connected component.

Two implementations of the same definition: scipy.ndimage on the host (what the reference runs), and - when a GPU and the HIP library are
there and the input is a float64 3-D volume, i.e. the output of `patch_wise_prediction` - the device kernels of csrc/postprocess.hip
(`fmri_correlate1d_f64`, `fmri_threshold_f64`, `fmri_fill_holes_step`, `fmri_largest_component_step`): the gaussian sums in scipy's own
order in fp64, so the two produce the same mask voxel for voxel (tests/test_gpu_postprocess.py).  `device=` forces one or the other."""
import numpy as np
from scipy import ndimage


def get_main_connected_component(data):
    """boolean mask of the largest 6-connected component of `data` (all False when there is none)"""
    components, count = ndimage.label(data)
    if count == 0:
        return np.zeros(components.shape, dtype=bool)          # the reference raises on the argmax of an empty list here
    voxels_per_component = np.bincount(components.ravel(), minlength=count + 1)[1:]
    return components == 1 + int(np.argmax(voxels_per_component))


def _device_ok(pred):
    if not (isinstance(pred, np.ndarray) and pred.ndim == 3 and pred.dtype == np.float64):
        return False
    try:
        import torch
        from fmri_hip._lib import lib
        if not torch.cuda.is_available():
            return False
        lib()
        return True
    except Exception:
        return False


def postprocess_prediction(pred, gaussian_std=1, threshold=0.5, fill_holes=True, connected_component=True, device=None):
    """device=None: the device path when it applies (see the module header), else scipy; True / False force one"""
    if device is None:
        device = _device_ok(pred)
    if device:
        return _postprocess_on_device(pred, gaussian_std, threshold, fill_holes, connected_component)
    mask = ndimage.gaussian_filter(pred, gaussian_std) > threshold
    steps = []
    if fill_holes:
        steps.append(ndimage.binary_fill_holes)
    if connected_component:
        steps.append(get_main_connected_component)
    for step in steps:
        mask = step(mask)
    return mask


def _postprocess_on_device(pred, gaussian_std, threshold, fill_holes, connected_component):
    import torch
    from fmri_hip import ops
    vol = pred if isinstance(pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float64)).cuda()
    mask = ops.threshold_f64(ops.gaussian_filter_f64(vol, gaussian_std), threshold)
    if fill_holes:
        mask = ops.binary_fill_holes_u8(mask)
    if connected_component:
        mask = ops.largest_component_u8(mask)
    return mask.cpu().numpy().astype(bool)
