"""Oracle sampler / augmentation / TTA vs outputs of the reference's augment.py, generator.py, utils.py and prediction.py
(tests/golden/augment_golden.*, made by tests/golden/make_augment_fixtures.py)."""
import json
import os
import random

import numpy as np
import pytest
import scipy.ndimage

from oracle import augment_oracle as A
from oracle import tiler_oracle as T


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "augment_golden.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, "augment_golden.npz"))


def synth_volumes(seed, shapes):
    """the same synthetic cases the fixture script fed to the reference"""
    rs = np.random.RandomState(seed)
    vols, truths = [], []
    for s in shapes:
        v = scipy.ndimage.gaussian_filter(rs.randn(*s), 1.5) * 4.0 + 0.3 * rs.randn(*s)
        t = (scipy.ndimage.gaussian_filter(rs.randn(*s), 2.0) > 0.02).astype(np.uint8)
        vols.append(v.astype(np.float64))
        truths.append(t)
    return vols, truths


def test_distort_affines(gold):
    meta, arr = gold
    for k, c in enumerate(meta["affine_cases"]):
        got = A.distort_affine(c["shape"], c["flip"], c["scale"], None if c["rotate"] is None else np.array(c["rotate"]),
                               None if c["translate"] is None else np.array(c["translate"]))
        np.testing.assert_allclose(got, arr["affine_%d" % k], rtol=0, atol=1e-12)


def test_interpolate_affine_range(gold):
    meta, arr = gold
    vol, lab = arr["interp_vol"], arr["interp_lab"]
    for k, c in enumerate(meta["interp_cases"]):
        Aff = arr["interp_A_%d" % k]
        ranges = [tuple(r) for r in c["ranges"]]
        np.testing.assert_allclose(A.interpolate_affine_range(vol, Aff, ranges, order=1, cval=c["cval1"]), arr["interp_o1_%d" % k],
                                   rtol=0, atol=1e-12)
        np.testing.assert_array_equal(A.interpolate_affine_range(lab, Aff, ranges, order=0, cval=0), arr["interp_o0_%d" % k])


def test_generator_batches(gold):
    meta, arr = gold
    for c in meta["generator_cases"]:
        vols, truths = synth_volumes(c["seed"], [tuple(s) for s in c["shapes"]])
        kw = dict(c["kwargs"])
        df = A.DataFileDummy(vols, truths, 3, tuple(c["patch"]))
        np.random.seed(c["seed"])
        random.seed(c["seed"])
        g = A.data_generator(df, list(range(len(vols))), kw["batch_size"], tuple(c["patch"]), augment=kw.get("augment"),
                             skip_blank=kw["skip_blank"], truth_index=kw["truth_index"], truth_size=kw["truth_size"],
                             prev_truth_index=kw.get("prev_truth_index"), prev_truth_size=kw.get("prev_truth_size"), is3d=kw["is3d"])
        for b in range(c["n_batches"]):
            x, y = next(g)
            gx, gy = arr["%s_x%d" % (c["name"], b)], arr["%s_y%d" % (c["name"], b)]
            assert x.shape == gx.shape and y.shape == gy.shape, c["name"]
            np.testing.assert_allclose(x, gx, rtol=0, atol=1e-10, err_msg=c["name"])
            np.testing.assert_array_equal(y, gy, err_msg=c["name"])


def test_permutations(gold):
    meta, arr = gold
    keys = sorted(A.generate_permutation_keys())
    assert [[list(k[0])] + list(k[1:]) for k in keys] == meta["permutation_keys"] and len(keys) == 48
    cube = arr["perm_in"]
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(A.permute_data(cube, k), arr["perm_out"][i])
        np.testing.assert_array_equal(A.reverse_permute_data(A.permute_data(cube, k), k), arr["perm_back"][i])


class FakeModel3D:
    def __init__(self, patch, n_out=1):
        self.patch = tuple(patch)
        self.output_shape = (None, n_out) + self.patch
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


def test_tta(gold):
    meta, arr = gold
    vol = arr["tta_vol"]
    fm = FakeModel3D(meta["tta"]["flips_patch"])

    def pw(v):
        return T.patch_wise_prediction(fm, v, tuple(meta["tta"]["flips_patch"]), overlap_factor=meta["tta"]["overlap_factor"])

    flips = A.predict_flips(pw, vol)
    np.testing.assert_allclose(np.stack(flips), arr["tta_flips"], rtol=0, atol=1e-12)
    cube_in = arr["tta_perm_in"]
    got = np.asarray([A.predict_with_permutations(Cube().predict, cube_in[b]) for b in range(cube_in.shape[0])])
    np.testing.assert_allclose(got, arr["tta_perm_out"], rtol=0, atol=1e-12)
    for seed in meta["tta"]["augment_seeds"]:
        np.random.seed(seed)
        np.testing.assert_allclose(A.predict_augment(pw, vol, num_augments=1), arr["tta_augment_%d" % seed], rtol=0, atol=1e-10)


# ------------------------------------------------------------------------------------------------ scikit-image / scikit-learn pins
def _sk(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "skimage_golden.npz"))


def _cases(z, prefix):
    n = 0
    while "%s_%d" % (prefix, n) in z.files:
        key = "%s_%d" % (prefix, n)
        yield key, z["in_" + str(z[key + "_in"])], z[key + "_args"], z[key]
        n += 1


def test_contrast_augment_matches_skimage_rescale_intensity(golden_dir):
    """reference contrast_augment over scikit-image 0.18.3 (the fixture) vs the restatement, incl. a window outside the data range, an
    all-negative volume and a nearly empty window"""
    z = _sk(golden_dir)
    seen = 0
    for key, vol, (lo, hi), want in _cases(z, "contrast"):
        got = A.contrast_augment(vol, lo, hi)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), key
        seen += 1
    assert seen == 5


def test_noise_augmentations_match_skimage_random_noise(golden_dir):
    """add_gaussian_noise / add_speckle_noise: the fixture was drawn from numpy's global generator seeded right before the call; skimage
    draws normal(0, sqrt(sigma^2), shape) once, which the legacy RandomState of any numpy reproduces"""
    z = _sk(golden_dir)
    for kind, fn in (("gaussian", A.add_gaussian_noise), ("speckle", A.add_speckle_noise)):
        seen = 0
        for key, vol, (sigma, seed), want in _cases(z, kind):
            draw = np.random.RandomState(int(seed)).normal(0.0, (sigma ** 2) ** 0.5, vol.shape)
            got = fn(vol, sigma, draw / sigma)
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), key
            seen += 1
        assert seen == 4


def test_shot_noise_matches_skimage_poisson(golden_dir):
    z = _sk(golden_dir)
    seen = 0
    for key, vol, (seed,), want in _cases(z, "shot"):
        rs = np.random.RandomState(int(seed))
        got = A.shot_noise(vol, poisson=rs.poisson)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), key
        seen += 1
    assert seen == 3


def test_gaussian_filter_matches_skimage_filters_gaussian(golden_dir):
    z = _sk(golden_dir)
    seen = 0
    for key, vol, (sigma,), want in _cases(z, "gfilter"):
        got = A.apply_gaussian_filter(vol, sigma)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), key
        seen += 1
    assert seen == 5
    # skimage's RGB guess: a 3-D array with 3 planes along the last axis is not smoothed along it
    vol = np.random.RandomState(0).rand(6, 5, 3)
    got = A.apply_gaussian_filter(vol, 1.0)
    from scipy import ndimage
    assert np.allclose(got, ndimage.gaussian_filter(vol, [1.0, 1.0, 0.0], mode="nearest", truncate=4.0))
