"""Host side of the augmentation / TTA rows (fetal_net.augment, fetal_net.prediction TTA entry points, fetal_net.postprocess) against
the reference's outputs in tests/golden/augment_golden.* - no GPU needed: the fake models take the duck-typed host-tile path."""
import json
import os
import random

import numpy as np
import pytest
from scipy import ndimage

from fetal_net import augment as AU
from fetal_net import postprocess as PP
from fetal_net import prediction as P
from oracle import augment_oracle as OA


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "augment_golden.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, "augment_golden.npz"))


def test_distort_image_matches_reference(gold):
    meta, arr = gold
    for k, c in enumerate(meta["affine_cases"]):
        _, A = AU.distort_image(np.zeros(c["shape"]), np.eye(4), flip_axis=None if c["flip"] is None else np.array(c["flip"]),
                                scale_factor=c["scale"], rotate_factor=None if c["rotate"] is None else np.array(c["rotate"]),
                                translate_factor=None if c["translate"] is None else np.array(c["translate"]))
        np.testing.assert_allclose(A, arr["affine_%d" % k], rtol=0, atol=1e-12)


@pytest.mark.parametrize("cfg", [
    {"flip": [0.5, 0.5, 0.5], "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "translate": (15, 15, 7)},
    {"flip": [0.5, 0.5, 0.5], "scale": (0.15, 0.15, 0), "iso_scale": {"max": 1.2}, "rotate": (5, 5, 45), "translate": (2, 2, 1),
     "contrast": {"min_factor": 0.2, "max_factor": 0.1}, "intensity_multiplication": (0.8, 1.2), "poisson_noise": 1,
     "gaussian_filter": {"prob": 0.3, "max_sigma": 1}, "elastic_transform": {"alpha": 5, "sigma": 10},
     "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05},
     "coarse_dropout": {"rate": 0.2, "size_percent": [0.1, 0.3], "per_channel": True}},
    {"rotate": None, "flip": None},
])
def test_random_draws_follow_the_reference_order(cfg):
    """the oracle's draw order is pinned to the reference by test_oracle_augment.test_generator_batches; the product must consume the
    generators identically, including the draws of augmenters it does not apply"""
    for seed in (1, 2, 3):
        np.random.seed(seed)
        random.seed(seed)
        a = AU.draw_augment_parameters(cfg, 3, -1.5, 7.0)
        tail_a = (np.random.random(), random.random())
        np.random.seed(seed)
        random.seed(seed)
        b = OA.draw_augment_params(cfg, 3, -1.5, 7.0)
        tail_b = (np.random.random(), random.random())
        assert tail_a == tail_b
        np.testing.assert_array_equal(np.asarray(a["scale_factor"]), np.asarray(b["scale"]))
        for ka, kb in [("rotate_factor", "rotate"), ("flip_axis", "flip"), ("translate_factor", "translate"), ("contrast", "contrast")]:
            if b[kb] is None:
                assert a[ka] is None
            else:
                np.testing.assert_array_equal(np.asarray(a[ka]), np.asarray(b[kb]))
        assert a["intensity_multiplication"] == b["intensity"]
        assert (a["apply_gaussian_noise"], a["apply_speckle_noise"]) == (b["gaussian_noise"], b["speckle_noise"])


def test_permutation_functions(gold):
    meta, arr = gold
    keys = sorted(AU.generate_permutation_keys())
    assert [[list(k[0])] + list(k[1:]) for k in keys] == meta["permutation_keys"]
    cube = arr["perm_in"]
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(AU.permute_data(cube, k), arr["perm_out"][i])
        np.testing.assert_array_equal(AU.reverse_permute_data(AU.permute_data(cube, k), k), arr["perm_back"][i])


class FakeModel3D:
    def __init__(self, patch):
        self.patch = tuple(patch)
        self.output_shape = (None, 1) + self.patch
        g = np.meshgrid(*[np.arange(s) for s in self.patch], indexing="ij")
        self.ramp = (g[0] * 1.0 + g[1] * 0.5 + g[2] * 0.25) / float(sum(self.patch))

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.stack([np.tanh(0.5 * x[:, 0]) + 0.01 * self.ramp[None]], axis=1)


class Cube:
    output_shape = (None, 1, 8, 8, 8)

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        g = np.meshgrid(*[np.arange(8)] * 3, indexing="ij")
        return np.tanh(0.5 * x) + 0.01 * (g[0] + 0.5 * g[1] + 0.25 * g[2])[None, None] / 24.0


def test_tta_entry_points_match_reference(gold):
    meta, arr = gold
    vol = arr["tta_vol"]
    patch = meta["tta"]["flips_patch"]
    fm = FakeModel3D(patch)
    flips = P.predict_flips(vol, fm, meta["tta"]["overlap_factor"], {"patch_shape": patch[:2], "patch_depth": patch[2]})
    assert len(flips) == 8
    np.testing.assert_allclose(np.stack(flips), arr["tta_flips"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(P.predict(Cube(), arr["tta_perm_in"], permute=True), arr["tta_perm_out"], rtol=0, atol=1e-12)
    for seed in meta["tta"]["augment_seeds"]:
        np.random.seed(seed)
        got = P.predict_augment(vol, fm, overlap_factor=meta["tta"]["overlap_factor"], patch_shape=tuple(patch), num_augments=1)
        np.testing.assert_allclose(got, arr["tta_augment_%d" % seed], rtol=0, atol=1e-10)


def test_postprocess_prediction():
    rs = np.random.RandomState(5)
    pred = ndimage.gaussian_filter(rs.rand(24, 24, 12), 2.0)
    pred = (pred - pred.min()) / (pred.max() - pred.min())
    pred[2:4, 2:4, 2:4] = 1.0                                 # a small island that the largest-component step must drop
    got = PP.postprocess_prediction(pred)
    want = ndimage.binary_fill_holes(ndimage.gaussian_filter(pred, 1) > 0.5)
    lab, n = ndimage.label(want)
    sizes = [np.sum(lab == i) for i in range(1, n + 1)]
    want = lab == (int(np.argmax(sizes)) + 1)
    assert n >= 2 and got.dtype == bool
    np.testing.assert_array_equal(got, want)
    assert not PP.postprocess_prediction(np.zeros((8, 8, 8))).any()
