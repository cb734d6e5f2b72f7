"""Whole-volume prediction pipeline of the reference's production entry point (reference prod/predict_nifti2.py:25-178), as library
functions: pre-processing -> patch-wise prediction on the MI355X (optionally with test-time augmentation) -> un-pad / re-scale ->
optional SECOND stage: threshold + clean the first mask, crop the padded bounding box from the ORIGINAL volume, predict it with a
second (higher resolution) model and paste the result back.

Only the model work runs on the device (`patch_wise_prediction`, `predict_flips`, `predict_augment`); zoom, windowing, normalisation,
the connected-component clean-up and the crop / paste are once-per-volume host passes on numpy / scipy, as in the reference.  File I/O is
left to the caller (`fetal_net.utils.nifti` reads and writes NIfTI-1); argument names follow the reference's `main()`.
"""
import numpy as np
from scipy import ndimage

from .postprocess import postprocess_prediction
from .prediction import patch_wise_prediction, predict_augment, predict_flips
from .utils.cut_relevant_areas import check_bounding_box, find_bounding_box

ROI_PADDING = (16, 16, 8)          # reference predict_nifti2.py:31


def window_intensities_data(data, min_percent=1, max_percent=99, out_min=0.0, out_max=255.0):
    """SimpleITK IntensityWindowing(image, p_lo, p_hi) restated (reference fetal/preprocess.py:50-55): the [p_lo, p_hi] percentile window
    is mapped linearly onto [0, 255], values outside it are clamped"""
    data = np.asarray(data, dtype=np.float64)
    lo, hi = np.percentile(data, min_percent), np.percentile(data, max_percent)
    if hi == lo:
        return np.full(data.shape, out_min)
    return (np.clip(data, lo, hi) - lo) * ((out_max - out_min) / (hi - lo)) + out_min


def normalize_data(data, mean, std):
    """reference fetal_net/normalize.py:66-69"""
    return (np.asarray(data, dtype=np.float64) - mean) / std


def preproc_and_norm(data, preprocess_method=None, norm_params=None, scale=None, preproc=None):
    """reference predict_nifti2.py:56-73.  `preproc`: a callable (the reference looks a name up in its fetal_net.preprocess module)"""
    if preprocess_method is not None:
        if preprocess_method != 'window_1_99':
            raise Exception('Unknown preprocess: {}'.format(preprocess_method))
        data = window_intensities_data(data)
    if scale is not None:
        data = ndimage.zoom(data, scale)
    if preproc is not None:
        if not callable(preproc):
            raise TypeError("preproc must be a callable here (the reference's name lookup in fetal_net.preprocess is not part of this package)")
        data = preproc(data)
    if norm_params is not None and any(norm_params.values()):
        data = normalize_data(data, mean=norm_params['mean'], std=norm_params['std'])
    return data


def get_prediction(data, model, augment, num_augments, return_all_preds, overlap_factor, config):
    """reference predict_nifti2.py:76-95: plain / 'flip' (8 variants) / 'all' (num_augments random variants); the variants are merged
    by their median unless return_all_preds"""
    patch_shape = list(config["patch_shape"]) + [config["patch_depth"]]
    if augment is not None:
        if augment == 'all':
            prediction = predict_augment(data, model=model, overlap_factor=overlap_factor, num_augments=num_augments, patch_shape=patch_shape)
        elif augment == 'flip':
            prediction = np.stack(predict_flips(data, model=model, overlap_factor=overlap_factor, config=config))
        else:
            raise ValueError("Unknown augmentation {}".format(augment))
        if not return_all_preds:
            prediction = np.median(prediction, axis=0)
    else:
        prediction = patch_wise_prediction(model=model, data=np.expand_dims(data, 0), overlap_factor=overlap_factor, patch_shape=patch_shape)
    return np.asarray(prediction).squeeze()


def secondary_prediction(mask, vol, config2, model2, preprocess_method2=None, norm_params2=None, overlap_factor=0.9, augment2=None,
                         num_augment=32, return_all_preds=False, padding=ROI_PADDING):
    """Second stage (reference predict_nifti2.py:25-53): bounding box of the first-stage mask, grown by `padding` and clipped to the
    volume; that box of the ORIGINAL volume is pre-processed with the second model's parameters and predicted; the result is zero-padded
    back to the volume's shape.  `model2`: a loaded model (the reference loads `get_last_model_path(model2_path)` here)."""
    vol = np.asarray(vol)
    bbox_start, bbox_end = find_bounding_box(mask)
    check_bounding_box(mask, bbox_start, bbox_end)
    if padding is not None:
        bbox_start = np.maximum(bbox_start - np.asarray(padding), 0)
        bbox_end = np.minimum(bbox_end + np.asarray(padding), np.asarray(mask).shape)
    data = vol.astype(np.float64)[bbox_start[0]:bbox_end[0], bbox_start[1]:bbox_end[1], bbox_start[2]:bbox_end[2]]
    data = preproc_and_norm(data, preprocess_method2, norm_params2)
    prediction = get_prediction(data, model2, augment=augment2, num_augments=num_augment, return_all_preds=return_all_preds,
                                overlap_factor=overlap_factor, config=config2)
    pad_back = list(zip(bbox_start, np.array(vol.shape) - bbox_end))
    if return_all_preds:
        pad_back = [(0, 0)] + pad_back
    return np.pad(prediction, pad_back, mode='constant', constant_values=0)


def predict_volume(data, model, config, overlap_factor=0.9, preprocess_method=None, norm_params=None, augment=None, num_augment=0,
                   model2=None, config2=None, preprocess_method2=None, norm_params2=None, augment2=None, num_augment2=0,
                   z_scale=None, xy_scale=None, return_all_preds=False):
    """The body of the reference's `main()` (predict_nifti2.py:98-160) on arrays: `data` = the volume as read from the NIfTI file.
    Returns a dict: 'data' (the pre-processed volume the first model saw, before its 3-voxel padding), 'prediction' (first stage, back at
    the input resolution) and, with model2 / config2, 'prediction_roi' (second stage on the padded bounding box, volume-sized)."""
    original = np.asarray(data, dtype=np.float64).squeeze()
    data = original
    z_scale = 1.0 if z_scale is None else z_scale
    xy_scale = 1.0 if xy_scale is None else xy_scale
    if z_scale != 1.0 or xy_scale != 1.0:
        data = ndimage.zoom(data, [xy_scale, xy_scale, z_scale])
    data = preproc_and_norm(data, preprocess_method, norm_params, scale=config.get('scale_data', None), preproc=config.get('preproc', None))
    out = {"data": data}
    padded = np.pad(data, 3, 'constant', constant_values=data.min())
    prediction = get_prediction(data=padded, model=model, augment=augment, num_augments=num_augment, return_all_preds=return_all_preds,
                                overlap_factor=overlap_factor, config=config)
    prediction = prediction[..., 3:-3, 3:-3, 3:-3]
    if config.get('scale_data', None) is not None:                      # back to the size before the model-specific scaling
        prediction = ndimage.zoom(prediction.squeeze(), np.divide([1, 1, 1], config['scale_data']), order=0)[..., np.newaxis]
    if z_scale != 1.0 or xy_scale != 1.0:
        prediction = ndimage.zoom(prediction.squeeze(), [1.0 / xy_scale, 1.0 / xy_scale, 1.0 / z_scale], order=1)[..., np.newaxis]
    out["prediction"] = prediction
    if config2 is not None:
        if model2 is None:
            raise ValueError("config2 given without model2")
        mask = postprocess_prediction(prediction.squeeze(), gaussian_std=0.5, threshold=0.5)
        out["mask"] = mask
        out["prediction_roi"] = secondary_prediction(mask, vol=original, config2=config2, model2=model2, preprocess_method2=preprocess_method2,
                                                     norm_params2=norm_params2, overlap_factor=overlap_factor, augment2=augment2,
                                                     num_augment=num_augment2, return_all_preds=return_all_preds)
    return out
