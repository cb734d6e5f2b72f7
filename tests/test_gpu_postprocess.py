"""Device post-processing (csrc/postprocess.hip) against scipy.ndimage - the reference's own implementation of
fetal_net/postprocess.py:7-19 - on seeded volumes: the gaussian bit for bit, the masks voxel for voxel; edge cases: holes touching the
border (not holes), nested holes, several components of equal size (tie -> the first in scan order), an empty mask, a full mask."""
import numpy as np
import pytest
import torch
from scipy import ndimage

pytestmark = pytest.mark.gpu


def _blobs(shape, seed, sigma=3.0, q=0.6):
    rs = np.random.RandomState(seed)
    f = ndimage.gaussian_filter(rs.randn(*shape), sigma)
    return (f > np.quantile(f, q)).astype(np.uint8)


@pytest.mark.parametrize("shape", [(40, 48, 56), (17, 33, 9), (64, 64, 64)])
def test_gaussian_filter_bit_exact(shape):
    from fmri_hip import ops
    rs = np.random.RandomState(1)
    v = rs.rand(*shape)
    for sigma in (1, 0.5, 2.3, (1.0, 0.0, 2.0)):
        got = ops.gaussian_filter_f64(torch.from_numpy(v).cuda(), sigma).cpu().numpy()
        ref = ndimage.gaussian_filter(v, sigma)
        assert np.array_equal(got, ref), (shape, sigma, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fill_holes_and_largest_component_equal_scipy(seed):
    from fmri_hip import ops
    from fetal_net.postprocess import get_main_connected_component
    m = _blobs((48, 56, 40), seed)
    md = torch.from_numpy(m).cuda()
    filled = ops.binary_fill_holes_u8(md).cpu().numpy().astype(bool)
    assert np.array_equal(filled, ndimage.binary_fill_holes(m))
    big = ops.largest_component_u8(md).cpu().numpy().astype(bool)
    assert np.array_equal(big, get_main_connected_component(m))


def test_edge_cases():
    from fmri_hip import ops
    from fetal_net.postprocess import get_main_connected_component
    shape = (20, 24, 28)
    m = np.zeros(shape, np.uint8)
    m[4:16, 4:20, 4:24] = 1
    m[7:13, 8:16, 8:20] = 0                       # a hole ...
    m[9:11, 10:14, 10:18] = 1                     # ... with an island inside (its own component, filled along with the hole)
    m[4:16, 10:12, 0:8] = 1                       # an arm to the border
    m[10, 11, 0:12] = 0                           # a tunnel from the border into the hole: not a hole any more
    for vol in (m, np.zeros(shape, np.uint8), np.ones(shape, np.uint8)):
        d = torch.from_numpy(vol).cuda()
        assert np.array_equal(ops.binary_fill_holes_u8(d).cpu().numpy().astype(bool), ndimage.binary_fill_holes(vol))
        assert np.array_equal(ops.largest_component_u8(d).cpu().numpy().astype(bool), get_main_connected_component(vol))
    t = np.zeros(shape, np.uint8)                 # three components of 8 voxels each: scipy's argmax takes the first in scan order
    t[10:12, 10:12, 20:22] = 1
    t[2:4, 2:4, 2:4] = 1
    t[15:17, 3:5, 7:9] = 1
    got = ops.largest_component_u8(torch.from_numpy(t).cuda()).cpu().numpy().astype(bool)
    assert np.array_equal(got, get_main_connected_component(t)) and got[2, 2, 2] and got.sum() == 8
    s = np.zeros((9, 40, 40), np.uint8)           # a long snake: many propagation steps
    s[4, 2:38:4, 2:38] = 1
    s[4, 2:38, 2] = 1
    assert np.array_equal(ops.largest_component_u8(torch.from_numpy(s).cuda()).cpu().numpy().astype(bool), get_main_connected_component(s))


@pytest.mark.parametrize("kw", [dict(), dict(gaussian_std=0.5), dict(fill_holes=False), dict(connected_component=False, threshold=0.4)])
def test_postprocess_prediction_device_equals_host(kw):
    """the reference entry point on a volume-sized probability map (160x256x256 = BASELINE configs[4]) - device path == scipy path"""
    from fetal_net.postprocess import postprocess_prediction
    rs = np.random.RandomState(5)
    shape = (160, 256, 256) if not kw else (64, 96, 80)
    p = ndimage.gaussian_filter(rs.rand(*shape), 2.0)
    p = (p - p.min()) / (p.max() - p.min())
    dev = postprocess_prediction(p, device=True, **kw)
    host = postprocess_prediction(p, device=False, **kw)
    assert dev.dtype == bool and dev.shape == host.shape
    assert np.array_equal(dev, host)
    assert np.array_equal(postprocess_prediction(p, **kw), host)          # the default picks the device path here and agrees
