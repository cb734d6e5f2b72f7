"""Whole-volume, optionally two-stage prediction - the behaviour of the reference's production entry point
(reference prod/predict_nifti2.py:25-160) - organised as data instead of as a script:

    Stage          what one model needs: the model, its patch geometry, intensity preparation, test-time augmentation, tile overlap
    Resampling     a geometric change of the volume together with its inverse for the prediction (zoom back, crop back)
    VolumePipeline a first Stage over the whole volume and, optionally, a second Stage over the padded bounding box of the first mask

Each resampling applied on the way in is pushed on a stack and undone in reverse on the way out, so the prediction always comes back on
the voxel grid of the input.  Only the model work runs on the device (the tile loop behind `patch_wise_prediction`, the TTA variants,
the mask clean-up when `postprocess_prediction` takes its device path); windowing, normalisation, zoom, crop and paste are
once-per-volume host passes.  File I/O is the caller's (`fetal_net.utils.nifti`).  `predict_volume(...)` keeps the keyword surface of
the reference's `main()` for callers that want the one-call form.
"""
import numpy as np
from scipy import ndimage

from .postprocess import postprocess_prediction
from .prediction import patch_wise_prediction, predict_augment, predict_flips
from .utils.cut_relevant_areas import check_bounding_box, find_bounding_box

ROI_PADDING = (16, 16, 8)          # margin around the first-stage mask (reference predict_nifti2.py:31)
CONTEXT_MARGIN = 3                 # voxels of minimum-valued border the first model sees around the volume (reference :131)


# ------------------------------------------------------------------------------------------------------------ intensity preparation
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


INTENSITY_METHODS = {"window_1_99": window_intensities_data}


# ------------------------------------------------------------------------------------------------------------------ building blocks
class Resampling(object):
    """a change of the sampling grid and how a prediction made on the new grid returns to the old one"""

    def forward(self, vol):
        raise NotImplementedError

    def backward(self, pred):
        raise NotImplementedError


class Zoom(Resampling):
    """scipy zoom by per-axis factors; predictions return with `order_back` (0 for the model-specific scaling, 1 for the resolution change,
    as the reference does at predict_nifti2.py:139-143)"""

    def __init__(self, factors, order_back):
        self.factors = [float(f) for f in np.broadcast_to(factors, (3,))]
        self.order_back = order_back

    def forward(self, vol):
        return ndimage.zoom(vol, self.factors)

    def backward(self, pred):
        # leading axes (a stack of TTA variants) are left alone
        lead = [1.0] * (pred.ndim - 3)
        return ndimage.zoom(pred, lead + [1.0 / f for f in self.factors], order=self.order_back)


class Border(Resampling):
    """a constant border of the volume's minimum on the way in, cropped off the prediction on the way out"""

    def __init__(self, width):
        self.width = int(width)

    def forward(self, vol):
        return np.pad(vol, self.width, mode="constant", constant_values=vol.min())

    def backward(self, pred):
        w = self.width
        return pred[(Ellipsis,) + (slice(w, -w),) * 3] if w else pred


class Box(Resampling):
    """crop to [start, end) of a volume of `shape`; predictions are pasted back into zeros"""

    def __init__(self, start, end, shape):
        self.start, self.end, self.shape = np.asarray(start), np.asarray(end), tuple(shape)

    def forward(self, vol):
        return vol[tuple(slice(a, b) for a, b in zip(self.start, self.end))]

    def backward(self, pred):
        room = [(int(a), int(s - b)) for a, b, s in zip(self.start, self.end, self.shape)]
        return np.pad(pred, [(0, 0)] * (pred.ndim - 3) + room, mode="constant", constant_values=0)


class Stage(object):
    """One model of the pipeline and everything that belongs to it.  `config`: the model's experiment config (`patch_shape`,
    `patch_depth`, optional `scale_data`, optional callable `preproc`); `intensity`: None or a key of INTENSITY_METHODS; `norm`: None or
    {'mean', 'std'}; `augment`: None | 'flip' (the 8 flips) | 'all' (`n_augment` random variants)."""

    def __init__(self, model, config, intensity=None, norm=None, augment=None, n_augment=0, overlap=0.9):
        if intensity is not None and intensity not in INTENSITY_METHODS:
            raise Exception("Unknown preprocess: {}".format(intensity))
        if augment not in (None, "flip", "all"):
            raise ValueError("Unknown augmentation {}".format(augment))
        if config.get("preproc") is not None and not callable(config["preproc"]):
            raise TypeError("config['preproc'] must be a callable here (the reference looks a name up in its own fetal_net.preprocess)")
        self.model, self.config, self.intensity, self.norm = model, config, intensity, norm
        self.augment, self.n_augment, self.overlap = augment, n_augment, overlap

    @property
    def patch(self):
        return list(self.config["patch_shape"]) + [self.config["patch_depth"]]

    def intensities(self, vol, resamplings=None):
        """windowing -> the model-specific scaling (recorded in `resamplings`) -> the config's own hook -> z-scoring, in the reference's order"""
        if self.intensity is not None:
            vol = INTENSITY_METHODS[self.intensity](vol)
        if resamplings is not None and self.config.get("scale_data") is not None:
            step = Zoom(self.config["scale_data"], order_back=0)
            resamplings.append(step)
            vol = step.forward(vol)
        if resamplings is not None and self.config.get("preproc") is not None:
            vol = self.config["preproc"](vol)
        if self.norm is not None and any(self.norm.values()):
            vol = normalize_data(vol, mean=self.norm["mean"], std=self.norm["std"])
        return vol

    def infer(self, vol, keep_variants=False):
        """probabilities of `vol` [X,Y,Z]: tiled prediction, or the median over the stage's test-time augmentation variants"""
        if self.augment == "all":
            variants = predict_augment(vol, model=self.model, overlap_factor=self.overlap, num_augments=self.n_augment, patch_shape=self.patch)
        elif self.augment == "flip":
            variants = np.stack(predict_flips(vol, model=self.model, overlap_factor=self.overlap, config=self.config))
        else:
            return np.asarray(patch_wise_prediction(model=self.model, data=vol[np.newaxis], overlap_factor=self.overlap,
                                                    patch_shape=self.patch)).squeeze()
        return np.asarray(variants if keep_variants else np.median(variants, axis=0)).squeeze()


def _undo(pred, resamplings):
    for step in reversed(resamplings):
        pred = step.backward(pred)
    return pred


class VolumePipeline(object):
    """first Stage on the whole volume; optional second Stage on the region of interest the first one finds.  `resolution`: (xy, z) zoom
    applied before the first model and undone (order 1) on its prediction; `mask_options`: arguments of the clean-up that turns the first
    prediction into the region-of-interest mask."""

    def __init__(self, first, second=None, resolution=(1.0, 1.0), roi_padding=ROI_PADDING, mask_options=None):
        self.first, self.second = first, second
        self.resolution = tuple(1.0 if r is None else float(r) for r in resolution)
        self.roi_padding = roi_padding
        self.mask_options = dict(gaussian_std=0.5, threshold=0.5) if mask_options is None else dict(mask_options)

    def run_first(self, volume, keep_variants=False):
        """-> (what the first model saw before its border, its prediction on the input grid)"""
        steps = []
        vol = volume
        xy, z = self.resolution
        if (xy, z) != (1.0, 1.0):
            steps.append(Zoom([xy, xy, z], order_back=1))
            vol = steps[-1].forward(vol)
        vol = self.first.intensities(vol, steps)
        seen = vol
        steps.append(Border(CONTEXT_MARGIN))
        pred = self.first.infer(steps[-1].forward(vol), keep_variants)
        return seen, _undo(pred, steps)

    def region_of_interest(self, mask):
        lo, hi = find_bounding_box(mask)
        check_bounding_box(mask, lo, hi)
        if self.roi_padding is not None:
            lo = np.maximum(lo - np.asarray(self.roi_padding), 0)
            hi = np.minimum(hi + np.asarray(self.roi_padding), np.asarray(mask).shape)
        return Box(lo, hi, np.asarray(mask).shape)

    def run_second(self, volume, mask, keep_variants=False):
        """the second model on the box around `mask`, cut from the ORIGINAL volume and prepared with the second stage's own parameters
        (no scaling hook at this stage, as in the reference); zero outside the box"""
        box = self.region_of_interest(mask)
        roi = self.second.intensities(box.forward(np.asarray(volume, dtype=np.float64)))
        return box.backward(self.second.infer(roi, keep_variants))

    def __call__(self, volume, keep_variants=False):
        volume = np.asarray(volume, dtype=np.float64).squeeze()
        seen, pred = self.run_first(volume, keep_variants)
        out = {"data": seen, "prediction": pred}
        if self.second is not None:
            out["mask"] = postprocess_prediction(pred.squeeze(), **self.mask_options)
            out["prediction_roi"] = self.run_second(volume, out["mask"], keep_variants)
        return out


def predict_volume(data, model, config, overlap_factor=0.9, preprocess_method=None, norm_params=None, augment=None, num_augment=0,
                   model2=None, config2=None, preprocess_method2=None, norm_params2=None, augment2=None, num_augment2=0,
                   z_scale=None, xy_scale=None, return_all_preds=False):
    """One-call form with the argument names of the reference's `main()` (predict_nifti2.py:98-160), on arrays: `data` = the volume as read
    from the NIfTI file.  Returns a dict: 'data' (the prepared volume the first model saw, before its border), 'prediction' (first stage, on
    the input grid) and, with model2 / config2, 'mask' and 'prediction_roi' (second stage on the padded bounding box, volume-sized)."""
    if config2 is not None and model2 is None:
        raise ValueError("config2 given without model2")
    first = Stage(model, config, preprocess_method, norm_params, augment, num_augment, overlap_factor)
    second = None if config2 is None else Stage(model2, config2, preprocess_method2, norm_params2, augment2, num_augment2, overlap_factor)
    return VolumePipeline(first, second, resolution=(xy_scale, z_scale))(data, keep_variants=return_all_preds)


def secondary_prediction(mask, vol, config2, model2, preprocess_method2=None, norm_params2=None, overlap_factor=0.9, augment2=None,
                         num_augment=32, return_all_preds=False, padding=ROI_PADDING):
    """the second stage alone (reference predict_nifti2.py:25-53), for callers that already hold a first-stage mask"""
    stage = Stage(model2, config2, preprocess_method2, norm_params2, augment2, num_augment, overlap_factor)
    return VolumePipeline(None, stage, roi_padding=padding).run_second(np.asarray(vol), mask, keep_variants=return_all_preds)
