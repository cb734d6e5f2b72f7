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


# ---------------------------------------------------------------------------------------------- imgaug restatements (PARITY UNPINNED)
# imgaug is neither pinned by the reference (requirements.txt: "imgaug") nor installed anywhere here: no fixture can exist.  These tests
# hold the restatements (imgaug 0.4.0 as published) to their own definitions, written out differently.
def test_elastic_shift_maps_are_a_truncated_gaussian_blur_of_the_padded_noise():
    rs = np.random.RandomState(0)
    h, w, sigma, alpha = 20, 17, 3.0, 7.0
    k = A.elastic_ksize(sigma)
    assert (A.elastic_ksize(1.0), A.elastic_ksize(2.0), k, A.elastic_ksize(4.0), A.elastic_ksize(10.0)) == (5, 7, 9, 11, 27)
    noise = rs.rand(2, h + 2 * k, w + 2 * k) * 2 - 1
    dx, dy = A.elastic_shift_maps((h, w), alpha, sigma, noise)
    xs = np.arange(k) - (k - 1) / 2
    g = np.exp(-xs ** 2 / (2 * sigma ** 2))
    g /= g.sum()
    K = np.outer(g, g)
    r = k // 2
    for blk, got in ((noise[0], dx), (noise[1], dy)):
        for (i, j) in ((0, 0), (h - 1, w - 1), (7, 11)):
            ci, cj = i + k, j + k                                     # the kept part starts `k` in: the kernel (radius k // 2) never leaves the padded block
            want = (blk[ci - r:ci + r + 1, cj - r:cj + r + 1] * K).sum() * alpha
            assert got[i, j] == pytest.approx(want, abs=1e-12)
    # the reference's default (alpha <= 5, sigma 10) barely moves anything: smoothed uniform noise has a small standard deviation
    big = rs.rand(2, 128 + 54, 128 + 54) * 2 - 1
    ddx, _ = A.elastic_shift_maps((128, 128), 5.0, 10.0, big)
    assert np.abs(ddx).max() < 0.6


def test_elastic_apply_is_a_backward_warp_with_clamped_coordinates():
    rs = np.random.RandomState(1)
    img = rs.rand(9, 7, 2)
    zero = np.zeros((9, 7))
    assert np.array_equal(A.elastic_apply(img, zero, zero, 1), img) and np.array_equal(A.elastic_apply(img, zero, zero, 0), img)
    one = np.ones((9, 7))
    out = A.elastic_apply(img, one, zero, 1)                          # dx = 1: every voxel reads its left neighbour (x - dx), column 0 clamps
    assert np.allclose(out[:, 1:], img[:, :-1]) and np.allclose(out[:, 0], img[:, 0])
    half = np.full((9, 7), 0.5)
    out = A.elastic_apply(img, zero, half, 1)                         # dy = 0.5: the mean of a voxel and the one above it
    assert np.allclose(out[1:], 0.5 * (img[1:] + img[:-1]))
    lab = (rs.rand(9, 7, 1) > 0.5).astype(np.uint8)
    out0 = A.elastic_apply(lab, np.full((9, 7), 0.4), zero, 0)        # nearest: a shift of 0.4 rounds back to the voxel itself
    assert np.array_equal(out0, lab)


def test_coarse_dropout_enlarges_the_grid_by_nearest_neighbour_and_drops_to_the_minimum():
    rs = np.random.RandomState(2)
    data = rs.randn(10, 6, 3) * 2 + 5
    keep = np.ones((3, 2, 3), dtype=np.uint8)
    keep[1, 0, 2] = 0                                                  # grid cell (1, 0) of slice 2
    out = A.coarse_dropout(data, keep)
    rows = [i for i in range(10) if min(int(np.floor(i * 3 / 10)), 2) == 1]        # cv2.resize INTER_NEAREST: floor(i * hs / h)
    assert rows == [4, 5, 6]
    want = data.copy()
    want[4:7, 0:3, 2] = data.min()
    assert np.allclose(out, want, atol=1e-12)
    both = A.coarse_dropout(data, np.zeros((1, 1, 1), dtype=np.uint8))
    assert np.allclose(both, data.min())
    g = [A.coarse_dropout_grid((128, 128), [0.10, 0.30], np.random.RandomState(s)) for s in range(20)]
    assert set(v for pair in g for v in pair) == {12, 38}              # a list is a choice between its two values, per axis
    assert A.coarse_dropout_grid((5, 5), 0.01, rs) == (3, 3)          # min_size = 3 per side (imgaug 0.4.0 CoarseDropout default)


def test_piecewise_affine_restatement_on_a_2x2_grid():
    """scipy's Delaunay splits the four corners along the diagonal (0,0) - (h,w) for every extent (what the device kernel hard-codes); with unmoved
    points the warp is the identity; a pure translation of all four destinations is a shift with zeros entering"""
    from scipy.spatial import Delaunay
    for h, w in ((128, 128), (48, 64), (64, 48), (17, 300)):
        tess = Delaunay(np.array([[0, 0], [w, 0], [0, h], [w, h]], dtype=float))
        assert sorted(sorted(t) for t in tess.simplices.tolist()) == [[0, 1, 3], [0, 2, 3]]
    rs = np.random.RandomState(4)
    img = rs.rand(12, 9, 2) + 1
    src, dst = A.piecewise_affine_points((12, 9), np.zeros((4, 2)))
    assert np.array_equal(dst, [[0, 0], [0, 8], [11, 0], [11, 8]])     # clipped to the image: [0, h - 1] x [0, w - 1]
    out = A.piecewise_affine_apply(img, src, src, 1)
    assert np.allclose(out, img)
    shifted = src + np.array([2.0, 1.0])
    out = A.piecewise_affine_apply(img, src, shifted, 1)                # output (y, x) reads input (y + 2, x + 1)
    assert np.allclose(out[:-2, :-1], img[2:, 1:]) and np.allclose(out[-2:], 0) and np.allclose(out[:, -1:], 0)
