"""Two-stage whole-volume prediction (reference prod/predict_nifti2.py:25-160) with point-wise fake models, so that the expected result
can be written down independently: stage 1 on the windowed / normalised / 3-voxel-padded volume, mask clean-up, bounding box + (16,16,8)
padding on the ORIGINAL volume, stage 2 with its own normalisation, zero paste-back."""
import numpy as np
import pytest
from scipy import ndimage


class PointwiseModel:
    """predict(x) = sigmoid(gain * (x - offset)) voxel by voxel: patch-wise overlap-add of it equals the same map of the whole volume"""

    def __init__(self, patch, gain, offset):
        self.output_shape = (None, 1) + tuple(patch)
        self.gain, self.offset = gain, offset

    def predict(self, x):
        return 1.0 / (1.0 + np.exp(-self.gain * (np.asarray(x, dtype=np.float64) - self.offset)))


def _blob_volume(shape=(40, 44, 24), seed=0):
    rs = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float64)
    blob = np.exp(-(((g - np.array([22, 20, 13])) / np.array([6.0, 7.0, 4.0])) ** 2).sum(-1))
    return 100.0 + 400.0 * blob + 5.0 * rs.randn(*shape)


def test_window_intensities_matches_the_sitk_definition():
    from fetal_net.pipeline import window_intensities_data
    rs = np.random.RandomState(1)
    d = rs.randn(10, 12, 8) * 30 + 50
    w = window_intensities_data(d)
    lo, hi = np.percentile(d, 1), np.percentile(d, 99)
    assert w.min() == 0.0 and abs(w.max() - 255.0) < 1e-9
    inside = (d > lo) & (d < hi)
    np.testing.assert_allclose(w[inside], (d[inside] - lo) / (hi - lo) * 255.0, rtol=1e-12)
    assert np.all(w[d <= lo] == 0.0) and np.allclose(w[d >= hi], 255.0)


def test_bounding_box_helpers():
    from fetal_net.utils.cut_relevant_areas import check_bounding_box, find_bounding_box
    m = np.zeros((9, 8, 7), np.uint8)
    m[2:5, 3:4, 1:6] = 1
    s, e = find_bounding_box(m)
    assert list(s) == [2, 3, 1] and list(e) == [5, 4, 6] and check_bounding_box(m, s, e)
    assert not check_bounding_box(m, s, e - 1)
    with pytest.raises(ValueError):
        find_bounding_box(np.zeros((3, 3, 3)))


def test_two_stage_prediction_against_the_written_out_definition():
    from fetal_net.pipeline import predict_volume
    from fetal_net.postprocess import postprocess_prediction
    vol = _blob_volume()
    cfg1 = {"patch_shape": [16, 16], "patch_depth": 8}
    cfg2 = {"patch_shape": [16, 16], "patch_depth": 8}
    norm1 = {"mean": 120.0, "std": 60.0}
    norm2 = {"mean": 200.0, "std": 90.0}
    m1 = PointwiseModel((16, 16, 8), gain=2.0, offset=1.5)
    m2 = PointwiseModel((16, 16, 8), gain=3.0, offset=0.5)
    out = predict_volume(vol, m1, cfg1, overlap_factor=0.5, norm_params=norm1, model2=m2, config2=cfg2, norm_params2=norm2)
    # stage 1, written out
    want1 = m1.predict((vol - 120.0) / 60.0)
    assert out["prediction"].squeeze().shape == vol.shape
    np.testing.assert_allclose(out["prediction"].squeeze(), want1, rtol=0, atol=1e-9)
    mask = postprocess_prediction(want1, gaussian_std=0.5, threshold=0.5)
    assert mask.sum() > 100 and np.array_equal(out["mask"], mask)
    idx = np.array(np.nonzero(mask))
    start = np.maximum(idx.min(1) - [16, 16, 8], 0)
    end = np.minimum(idx.max(1) + 1 + [16, 16, 8], vol.shape)
    want2 = np.zeros(vol.shape)
    sl = tuple(slice(a, b) for a, b in zip(start, end))
    want2[sl] = m2.predict((vol[sl] - 200.0) / 90.0)
    assert out["prediction_roi"].shape == vol.shape
    np.testing.assert_allclose(out["prediction_roi"], want2, rtol=0, atol=1e-9)
    assert (start > 0).any() or (end < np.array(vol.shape)).any()      # the ROI really is a crop in this case


def test_flip_augmentation_median_and_resolution_round_trip():
    from fetal_net.pipeline import predict_volume
    vol = _blob_volume((32, 32, 16), seed=2)
    cfg = {"patch_shape": [16, 16], "patch_depth": 8}
    m = PointwiseModel((16, 16, 8), gain=1.0, offset=0.0)
    plain = predict_volume(vol, m, cfg, overlap_factor=0.5, norm_params={"mean": 100.0, "std": 100.0})["prediction"].squeeze()
    flips = predict_volume(vol, m, cfg, overlap_factor=0.5, norm_params={"mean": 100.0, "std": 100.0}, augment="flip")["prediction"].squeeze()
    np.testing.assert_allclose(flips, plain, atol=1e-9)               # a point-wise model commutes with the flips: median of 8 equal maps
    allp = predict_volume(vol, m, cfg, overlap_factor=0.5, norm_params={"mean": 100.0, "std": 100.0}, augment="flip", return_all_preds=True)
    assert allp["prediction"].shape[0] == 8
    scaled = predict_volume(vol, m, cfg, overlap_factor=0.5, norm_params={"mean": 100.0, "std": 100.0}, xy_scale=0.5, z_scale=1.0)
    assert scaled["prediction"].squeeze().shape == vol.shape          # predicted at half in-plane resolution, zoomed back (order 1)
    assert scaled["data"].shape == (16, 16, 16)
    with pytest.raises(Exception):
        predict_volume(vol, m, cfg, preprocess_method="nope")


def test_resampling_steps_return_predictions_to_the_input_grid():
    """the building blocks of VolumePipeline: every resampling undoes itself on a prediction (also on a stack of TTA variants)"""
    from fetal_net.pipeline import Border, Box, Stage, VolumePipeline, Zoom
    vol = _blob_volume((20, 24, 12), seed=3)
    b = Border(3)
    assert b.forward(vol).shape == (26, 30, 18) and b.forward(vol)[0, 0, 0] == vol.min()
    np.testing.assert_array_equal(b.backward(b.forward(vol)), vol)
    np.testing.assert_array_equal(b.backward(np.stack([b.forward(vol)] * 2)), np.stack([vol] * 2))
    box = Box([2, 3, 1], [10, 20, 9], vol.shape)
    back = box.backward(box.forward(vol))
    assert back.shape == vol.shape and np.array_equal(back[2:10, 3:20, 1:9], vol[2:10, 3:20, 1:9]) and back[0].sum() == 0
    assert box.backward(np.ones((3, 8, 17, 8))).shape == (3,) + vol.shape
    z = Zoom([0.5, 0.5, 1.0], order_back=1)
    assert z.forward(vol).shape == (10, 12, 12) and z.backward(z.forward(vol)).shape == vol.shape
    with pytest.raises(ValueError):
        Stage(None, {"patch_shape": [8, 8], "patch_depth": 4}, augment="rotate")
    with pytest.raises(TypeError):
        Stage(None, {"patch_shape": [8, 8], "patch_depth": 4, "preproc": "by_name"})
    # a region of interest that touches the volume's border is clipped, not shifted
    m = np.zeros(vol.shape, bool)
    m[0:4, 10:14, 5:8] = True
    roi = VolumePipeline(None, None).region_of_interest(m)
    assert list(roi.start) == [0, 0, 0] and list(roi.end) == [20, 24, 12]
    roi = VolumePipeline(None, None, roi_padding=(2, 2, 1)).region_of_interest(m)
    assert list(roi.start) == [0, 8, 4] and list(roi.end) == [6, 16, 9]
