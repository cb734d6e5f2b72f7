"""The reference's older prediction helpers (prediction.py:214-275; nothing in the reference calls them) - present under the same names with
the same results: checked against a line-by-line numpy evaluation of the reference's statements on small arrays, the image helpers through
the NIfTI round trip (fetal_net.utils.nifti)."""
import os
import pickle

import numpy as np
import pytest

from fetal_net import prediction as P
from fetal_net.utils.nifti import NiftiImage, get_image, load_nifti


class _Model:
    def predict(self, x):
        x = np.asarray(x, np.float32)
        return np.stack([x * 0.5, 1.0 - x * 0.5], axis=1)[np.newaxis].reshape((1, 2) + x.shape)


class _Root:
    pass


class _File:
    def __init__(self, data, affine):
        self.root = _Root()
        self.root.data, self.root.affine = data, affine


def test_every_legacy_name_of_the_reference_module_exists():
    for name in ("get_prediction_labels", "get_test_indices", "predict_from_data_file", "predict_and_get_image", "predict_from_data_file_and_get_image",
                 "predict_from_data_file_and_write_image", "prediction_to_image", "multi_class_prediction", "patch_wise_prediction",
                 "run_validation_case", "run_validation_cases", "predict", "predict_with_permutations", "predict_flips", "predict_augment"):
        assert callable(getattr(P, name)), name


def test_get_prediction_labels_follows_the_reference_statements():
    rs = np.random.RandomState(0)
    pred = rs.rand(2, 4, 4, 3, 5)                    # n_labels == x: the only shapes the reference's statements accept
    got = P.get_prediction_labels(pred, threshold=0.6, labels=(7, 9, 11))
    assert len(got) == 2
    for s in range(2):
        want = np.argmax(pred[s], axis=1)
        want[np.max(pred[s], axis=0) < 0.6] = 0
        for value in np.unique(want).tolist()[1:]:
            want[want == value] = (7, 9, 11)[value - 1]
        assert got[s].dtype == np.uint8 and np.array_equal(got[s], want.astype(np.uint8))
    with pytest.raises(IndexError):                  # any other shape fails in the reference's masking line, and here
        P.get_prediction_labels(rs.rand(1, 2, 4, 3, 5))


def test_get_test_indices_reads_the_pickled_split(tmp_path):
    f = tmp_path / "test_ids.pkl"
    with open(f, "wb") as h:
        pickle.dump([3, 1, 4], h)
    assert P.get_test_indices(str(f)) == [3, 1, 4]


def test_predict_from_data_file_helpers_and_image_round_trip(tmp_path):
    rs = np.random.RandomState(1)
    data = [rs.rand(6, 5, 4).astype(np.float32) for _ in range(2)]
    affine = np.diag([2.0, 3.0, 4.0, 1.0])
    f, m = _File(data, affine), _Model()
    assert np.array_equal(P.predict_from_data_file(m, f, 1), m.predict(data[1]))
    img = P.predict_from_data_file_and_get_image(m, f, 0)
    assert isinstance(img, NiftiImage) and np.array_equal(img.get_data(), m.predict(data[0])[0, 0]) and np.array_equal(img.affine, affine)
    out = str(tmp_path / "p.nii.gz")
    P.predict_from_data_file_and_write_image(m, f, 0, out)
    back, aff = load_nifti(out, return_affine=True)
    assert np.allclose(back, m.predict(data[0])[0, 0]) and np.allclose(aff, affine)


def test_prediction_to_image_and_multi_class_prediction():
    rs = np.random.RandomState(2)
    one = rs.rand(1, 1, 4, 5, 6)
    img = P.prediction_to_image(one)
    assert np.array_equal(img.get_data(), one[0]) and np.array_equal(img.affine, np.eye(4))
    lab = P.prediction_to_image(one, label_map=True, threshold=0.4, labels=(5,))
    assert lab.get_data().dtype == np.int8 and np.array_equal(lab.get_data(), np.where(one[0, 0] > 0.4, 5, 0))
    multi = rs.rand(2, 3, 4, 5, 6)
    imgs = P.prediction_to_image(multi)
    assert len(imgs) == 3 and all(np.array_equal(im.get_data(), multi[0, i]) for i, im in enumerate(imgs))
    assert [np.array_equal(a.get_data(), b.get_data()) for a, b in zip(imgs, P.multi_class_prediction(multi, np.eye(4)))] == [True] * 3
    with pytest.raises(RuntimeError):
        P.prediction_to_image(rs.rand(2, 1, 4, 5, 6))
    assert np.array_equal(get_image(one[0, 0]).affine, np.eye(4))
