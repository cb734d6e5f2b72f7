"""fetal_net.utils.patches: the reference's patch-grid helpers (utils/patches.py) - the cases of the reference's own test/test_predict.py on the
array layout its CURRENT code takes ((X, Y, Z, C); that test file still feeds the 3-D arrays of the upstream project and no longer runs
there), plus the grid and padding helpers against hand-computed values."""
import numpy as np

from fetal_net.utils.patches import (compute_patch_indices, fix_out_of_bound_patch_attempt, get_patch_from_3d_data, get_random_nd_index,
                                     get_set_of_patch_indices, reconstruct_from_patches)


def _patches(data, patch_shape, indices):
    """(X, Y, Z, C) data -> (px, py, pz, C) patches through the channels-first helper the prediction path uses"""
    return [np.moveaxis(get_patch_from_3d_data(np.moveaxis(data, -1, 0), patch_shape, i), 0, -1) for i in indices]


def test_compute_patch_indices_centres_the_grid():
    idx = compute_patch_indices((120, 144, 90), np.asarray((32, 32, 32)), 0)
    assert idx.shape == (4 * 5 * 3, 3) and idx.dtype.kind == "i"
    assert idx[0].tolist() == [-4, -8, -3] and idx[-1].tolist() == [92, 120, 61]          # overhang 8 / 16 / 6 split over both ends
    assert np.array_equal(compute_patch_indices((64, 64, 64), np.asarray((32, 32, 32)), 16, start=0)[:4], [[0, 0, 0], [0, 0, 16], [0, 0, 32], [0, 0, 48]])
    assert np.array_equal(get_set_of_patch_indices((0, 0, 0), (2, 4, 2), (1, 2, 1)), [[0, 0, 0], [0, 0, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0], [1, 0, 1], [1, 2, 0], [1, 2, 1]])


def test_reconstruct_from_patches_cases_of_the_reference_tests():
    shape = (120, 144, 90)
    data = np.arange(np.prod(shape), dtype=np.float64).reshape(shape + (1,))
    ps = np.asarray((32, 32, 32))
    idx = compute_patch_indices(shape, ps, 0)
    patches = _patches(data, ps, idx)
    assert np.array_equal(reconstruct_from_patches(patches, idx.copy(), data.shape), data)                    # test_reconstruct_from_patches
    both = patches + [p - 2 for p in patches]                                                                  # ..._with_overlapping_patches: the average
    assert np.array_equal(reconstruct_from_patches(both, np.concatenate([idx, idx]), data.shape), data - 1)
    shape2 = (72, 72, 72)
    for c in (1, 4):                                                                                           # ..._patches2 / ..._multiple_channels
        d2 = np.arange(np.prod(shape2) * c, dtype=np.float64).reshape(shape2 + (c,))
        ps2 = np.asarray((32, 32, 32))
        i_a, i_b = compute_patch_indices(shape2, ps2, 8), compute_patch_indices(shape2, ps2, 16)
        allp = _patches(d2, ps2, i_a) + _patches(d2, ps2, i_b)
        assert np.array_equal(reconstruct_from_patches(allp, np.concatenate([i_a, i_b]), d2.shape), d2)


def test_fix_out_of_bound_patch_attempt_equals_the_clamped_gather():
    rs = np.random.RandomState(0)
    data = rs.rand(2, 10, 12, 8)
    for index in ([-3, 0, 2], [6, 9, -1], [0, 0, 0], [7, 8, 5]):
        padded, fixed = fix_out_of_bound_patch_attempt(data, (6, 6, 6), np.asarray(index))
        want = padded[:, fixed[0]:fixed[0] + 6, fixed[1]:fixed[1] + 6, fixed[2]:fixed[2] + 6]
        assert np.array_equal(want, get_patch_from_3d_data(data, (6, 6, 6), index))
    np.random.seed(3)
    assert all(0 <= v <= m for v, m in zip(get_random_nd_index((4, 0, 7)), (4, 0, 7)))
