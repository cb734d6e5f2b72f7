"""The reference's small host modules under their names: fetal_net.preprocess (:5-27), fetal_net.normalize (:64-92), fetal_net.utils.utils (the
pickle / image / affine-resampling helpers) and utils.create_distance_masks - against scipy / numpy evaluations of the reference's statements"""
import numpy as np
import scipy.ndimage as ndi

from fetal_net import normalize, preprocess
from fetal_net.utils import create_distance_masks as CDM
from fetal_net.utils import utils as U
from fetal_net.utils.nifti import save_nifti


def test_preprocess_filters():
    d = np.random.RandomState(0).rand(12, 10, 8)
    assert np.array_equal(preprocess.laplace(d), ndi.laplace(d))
    assert np.array_equal(preprocess.grad(d), ndi.gaussian_gradient_magnitude(d, sigma=(1, 1, 1)))
    for f, g in ((preprocess.laplace_norm, ndi.laplace(d)), (preprocess.grad_norm, ndi.gaussian_gradient_magnitude(d, sigma=(1, 1, 1)))):
        got = f(d)
        assert np.isclose(got.min(), -1) and np.isclose(got.max(), 1) and np.allclose(got, -1 + 2 * (g - g.min()) / (g.max() - g.min()))


def test_normalize_module():
    rs = np.random.RandomState(1)
    vols = [rs.rand(6, 5, 4) * 10 + k for k in range(3)]
    d = vols[0].copy()
    out = normalize.normalize_data(d, 2.0, 4.0)
    assert out is d and np.allclose(d, (vols[0] - 2.0) / 4.0)
    store, mean, std = normalize.normalize_data_storage([v.copy() for v in vols])
    assert np.isclose(mean, np.mean([v.mean() for v in vols])) and np.isclose(std, np.mean([v.std() for v in vols]))
    assert np.allclose(store[2], (vols[2] - mean) / std)
    each, m2, s2 = normalize.normalize_data_storage_each([v.copy() for v in vols])
    assert m2 is None and s2 is None and all(abs(e.mean()) < 1e-12 and abs(e.std() - 1) < 1e-12 for e in each)


def test_utils_images_and_affine_resampling(tmp_path):
    rs = np.random.RandomState(2)
    vol = rs.rand(9, 8, 7).astype(np.float32)
    aff = np.array([[2.0, 0, 0, 1], [0, 3.0, 0, -2], [0, 0, 0.5, 4], [0, 0, 0, 1]])
    path = str(tmp_path / "v.nii.gz")
    save_nifti(vol, path, aff)
    img = U.read_img(path)
    assert np.array_equal(img.get_data(), vol) and np.allclose(img.affine, aff) and np.allclose(U.get_affine(path), aff)
    U.pickle_dump([1, 2, 3], str(tmp_path / "p.pkl"))
    assert U.pickle_load(str(tmp_path / "p.pkl")) == [1, 2, 3]
    A = np.eye(4)
    A[:3, :3] = [[0.9, 0.1, 0.0], [-0.1, 0.95, 0.05], [0.0, 0.02, 1.1]]
    A[:3, 3] = [0.5, -0.25, 0.3]
    ranges = [(1, 6), (0, 5), (2, 6)]
    got = U.interpolate_affine_range(vol.astype(np.float64), A, ranges, order=1, cval=-1.0)
    ii, jj, kk = np.meshgrid(*[np.arange(s, e) for s, e in ranges], indexing="ij")
    src = np.einsum("ab,b...->a...", A[:3, :3], np.stack([ii, jj, kk]).astype(np.float64)) + A[:3, 3].reshape(3, 1, 1, 1)
    assert got.shape == (5, 5, 4) and np.allclose(got, ndi.map_coordinates(vol.astype(np.float64), src, order=1, mode="constant", cval=-1.0))
    assert np.array_equal(U.interpolate_affine_range(vol, np.eye(4), [(0, 9), (0, 8), (0, 7)]), vol)


def test_distance_masks(tmp_path):
    m = np.zeros((10, 9, 6), np.uint8)
    m[3:7, 2:6, 1:4] = 1
    want = ndi.distance_transform_edt(m, sampling=(0.4, 0.4, 3.0)) + ndi.distance_transform_edt(1 - m, sampling=(0.4, 0.4, 3.0))
    assert np.array_equal(CDM.distance_mask(m), want) and (want > 0).all()
    case = tmp_path / "case1"
    case.mkdir()
    save_nifti(m, str(case / "truth.nii.gz"))
    out = CDM.create_distance_masks(str(tmp_path))
    assert len(out) == 1 and out[0].endswith("dists.nii.gz")
    from fetal_net.utils.nifti import load_nifti
    assert np.allclose(load_nifti(out[0]), want)
