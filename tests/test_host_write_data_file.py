"""fetal_net.data.write_data_to_file (reference data.py:42-74 + normalize.py:70-92): images in, a data file out that open_data_file and the
generators read - contents against a numpy evaluation of the reference's statements"""
import os

import numpy as np
import pytest
from scipy.ndimage import zoom

from fetal_net.data import open_data_file, write_data_to_file
from fetal_net.utils.nifti import save_nifti


def _subjects(tmp_path, with_mask):
    rs = np.random.RandomState(0)
    files, arrays = [], []
    for i, shape in enumerate([(12, 10, 8), (14, 10, 6)]):
        v = (rs.rand(*shape) * 100 + 20 * i).astype(np.float32)
        t = (rs.rand(*shape) > 0.7).astype(np.uint8)
        m = rs.rand(*shape).astype(np.float32)
        names = [str(tmp_path / ("s%d_%s.nii.gz" % (i, k))) for k in ("vol", "truth", "mask")]
        for a, n in zip((v, t, m), names):
            save_nifti(a, n)
        files.append(tuple(names[:3 if with_mask else 2]))
        arrays.append((v, t, m))
    return files, arrays


@pytest.mark.parametrize("normalize,scale,with_mask", [("all", None, False), ("each", None, True), (False, 0.5, False)])
def test_write_data_to_file_contents(tmp_path, normalize, scale, with_mask):
    files, arrays = _subjects(tmp_path, with_mask)
    out = str(tmp_path / "data.h5")
    got_file, (mean, std) = write_data_to_file(files, out, subject_ids=["a", "b"], normalize=normalize, scale=scale,
                                               preproc=(lambda d: d + 1.0) if scale else None)
    assert got_file == out and os.path.exists(out)
    want = []
    for v, t, m in arrays:
        v, t = v.astype(np.float64), t
        if scale is not None:
            v, t = (zoom(v.astype(np.float32), scale) + 1.0).astype(np.float64), zoom(t, scale, order=0)      # (the stored float32 is zoomed)
        want.append([v, t, m.astype(np.float64)])
    if normalize == "all":
        mu = np.mean([w[0].mean() for w in want])
        sd = np.mean([w[0].std() for w in want])
        assert np.isclose(mean, mu) and np.isclose(std, sd)
        for w in want:
            w[0] = (w[0] - mu) / sd
    elif normalize == "each":
        assert mean is None and std is None
        for w in want:
            w[0] = (w[0] - w[0].mean()) / w[0].std()
    else:
        assert mean is None and std is None
    f = open_data_file(out)
    try:
        assert [s.decode() for s in f.root.subject_ids] == ["a", "b"]
        for i, w in enumerate(want):
            d, t = np.asarray(f.root.data[i]), np.asarray(f.root.truth[i])
            assert d.dtype == np.float64 and t.dtype == np.uint8
            np.testing.assert_allclose(d, w[0], rtol=1e-12, atol=1e-12)
            assert np.array_equal(t, w[1])
            if with_mask:
                np.testing.assert_allclose(np.asarray(f.root.mask[i]), w[2], rtol=0, atol=0)
        assert (len(f.root.mask) == 2) if with_mask else (len(f.root.mask) == 0)
    finally:
        f.close()
