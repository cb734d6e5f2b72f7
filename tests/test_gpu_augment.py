"""GPU parity of the device-side patch sampler / intensity augmentation (csrc/augment.hip through the C ABI) and of the device
generator built on it, against scipy (map_coordinates), the numpy oracle and the reference's own generator output
(tests/golden/augment_golden.*).

Tolerances: label sampling (order 0) bit-exact; image sampling (order 1) 1e-5 absolute on O(1..10) data (the device evaluates the
source coordinates in fp64 like the host and interpolates in fp64, the volume itself is stored in fp32: 6e-8 relative);
elementwise intensity passes 1e-5 relative (fp32 arithmetic vs fp64 oracle)."""
import json
import os
import random

import numpy as np
import pytest
import scipy.ndimage
import torch

from oracle import augment_oracle as OA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from fmri_hip import ops as o
    return o


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "augment_golden.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, "augment_golden.npz"))


def _sample(ops, vol, A, ranges, order, cval, out_dtype):
    v = torch.from_numpy(np.ascontiguousarray(vol)).cuda()
    size = [e - s for s, e in ranges]
    out = torch.empty(size, device="cuda", dtype=out_dtype)
    ops.affine_sample(v, A, [s for s, _ in ranges], size, out, order=order, cval=cval)
    torch.cuda.synchronize()
    return out.float().cpu().numpy() if out.dtype == torch.bfloat16 else out.cpu().numpy()


def test_affine_sample_matches_reference_fixture(ops, gold):
    meta, arr = gold
    vol, lab = arr["interp_vol"].astype(np.float32), arr["interp_lab"]
    for k, c in enumerate(meta["interp_cases"]):
        A = arr["interp_A_%d" % k]
        ranges = [tuple(r) for r in c["ranges"]]
        got1 = _sample(ops, vol, A, ranges, 1, c["cval1"], torch.float32)
        np.testing.assert_allclose(got1, arr["interp_o1_%d" % k], rtol=0, atol=1e-5)
        got0 = _sample(ops, lab, A, ranges, 0, 0.0, torch.uint8)
        np.testing.assert_array_equal(got0, arr["interp_o0_%d" % k])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_affine_sample_random_affines_vs_scipy(ops, seed):
    rs = np.random.RandomState(seed)
    shape = (int(rs.randint(9, 40)), int(rs.randint(9, 40)), int(rs.randint(5, 24)))
    vol = rs.randn(*shape).astype(np.float32)
    lab = (rs.rand(*shape) > 0.5).astype(np.uint8)
    A = OA.distort_affine(shape, flip_axis=np.arange(3)[rs.rand(3) > 0.5], scale_factor=list(rs.normal(1, 0.2, 3)),
                          rotate_factor=np.deg2rad(rs.uniform(-90, 90, 3)), translate_factor=rs.uniform(-6, 6, 3))
    ranges = [(-3, shape[0] + 2), (1, shape[1] - 1), (0, shape[2] + 4)]
    want1 = OA.interpolate_affine_range(vol.astype(np.float64), A, ranges, order=1, cval=-3.25)
    want0 = OA.interpolate_affine_range(lab, A, ranges, order=0, cval=0)
    np.testing.assert_allclose(_sample(ops, vol, A, ranges, 1, -3.25, torch.float32), want1, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(_sample(ops, lab, A, ranges, 0, 0.0, torch.uint8), want0)
    # bf16 output = the fp32 result rounded once
    got_bf = _sample(ops, vol, A, ranges, 1, -3.25, torch.bfloat16)
    np.testing.assert_allclose(got_bf, want1, rtol=2 ** -8, atol=1e-5)


def test_affine_sample_strided_channel_slot(ops):
    rs = np.random.RandomState(4)
    vol = rs.randn(12, 10, 9).astype(np.float32)
    wide = torch.full((8, 8, 7), -99.0, device="cuda")
    ops.affine_sample(torch.from_numpy(vol).cuda(), np.eye(4), (2, 1, 3), (8, 8, 5), wide, order=1, cval=0.0, out_ld=7)
    torch.cuda.synchronize()
    w = wide.cpu().numpy()
    np.testing.assert_array_equal(w[..., :5], vol[2:10, 1:9, 3:8])
    assert np.all(w[..., 5:] == -99.0)


def test_intensity_passes_vs_oracle(ops):
    rs = np.random.RandomState(6)
    x = (rs.randn(16, 16, 8) * 3 + 1).astype(np.float32)
    t = torch.from_numpy(x.copy()).cuda()
    stats = torch.empty(2, device="cuda")
    ops.minmax(t, stats)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(stats.cpu().numpy(), np.array([x.min(), x.max()], dtype=np.float32))
    lo, hi, mult = float(x.min() + 0.7), float(x.max() - 1.1), 1.13
    ops.rescale_intensity(t, stats, True, lo, hi, mult)
    want = OA.contrast_augment(x.astype(np.float64), lo, hi) * mult
    np.testing.assert_allclose(t.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    for kind, fn in ((1, OA.add_speckle_noise), (0, OA.add_gaussian_noise)):
        cur = t.cpu().numpy().astype(np.float64)
        noise = rs.randn(x.size).astype(np.float32)
        ops.minmax(t, stats)
        ops.noise_augment(t, stats, torch.from_numpy(noise).cuda(), kind, 0.05)
        want = fn(cur, 0.05, noise.reshape(x.shape).astype(np.float64))
        np.testing.assert_allclose(t.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # degenerate contrast window (lo == hi) and the all-negative / constant inputs of the min-max keys
    c = torch.full((64,), -2.5, device="cuda")
    ops.minmax(c, stats)
    assert stats.cpu().tolist() == [-2.5, -2.5]


def synth_volumes(seed, shapes):
    rs = np.random.RandomState(seed)
    vols, truths = [], []
    for s in shapes:
        v = scipy.ndimage.gaussian_filter(rs.randn(*s), 1.5) * 4.0 + 0.3 * rs.randn(*s)
        t = (scipy.ndimage.gaussian_filter(rs.randn(*s), 2.0) > 0.02).astype(np.uint8)
        vols.append(v.astype(np.float64))
        truths.append(t)
    return vols, truths


class _Root:
    pass


class FakeDataFile:
    def __init__(self, vols, truths, masks=None):
        self.root = _Root()
        self.root.data, self.root.truth = vols, truths
        self.root.mask = masks if masks is not None else []
        self.root.subject_ids = [("s%d" % i).encode() for i in range(len(vols))]


def test_device_generator_reproduces_the_reference_batches(gold):
    """seeded like the fixture run of the reference's data_generator: same corners, same transformations, same batches"""
    from fetal_net.device_generator import device_data_generator
    meta, arr = gold
    for c in meta["generator_cases"]:
        vols, truths = synth_volumes(c["seed"], [tuple(s) for s in c["shapes"]])
        kw = dict(c["kwargs"])
        masks = None
        if c.get("masks"):          # distance masks of the mask-weighted loss -> ([x, masks], y) batches
            masks = [scipy.ndimage.distance_transform_edt(1 - t) + scipy.ndimage.distance_transform_edt(t) for t in truths]
        np.random.seed(c["seed"])
        random.seed(c["seed"])
        g = device_data_generator(FakeDataFile(vols, truths, masks), list(range(len(vols))), patch_shape=tuple(c["patch"]),
                                  shuffle_index_list=False, **kw)
        for b in range(c["n_batches"]):
            x, y = next(g)
            torch.cuda.synchronize()
            if masks is not None:
                x, m = x
                gm = arr["%s_m%d" % (c["name"], b)]
                assert m.is_cuda and tuple(m.shape) == gm.shape, c["name"]
                np.testing.assert_allclose(m.cpu().numpy(), gm, rtol=0, atol=1e-6, err_msg=c["name"] + " mask")
            gx, gy = arr["%s_x%d" % (c["name"], b)], arr["%s_y%d" % (c["name"], b)]
            assert tuple(x.shape) == gx.shape and tuple(y.shape) == gy.shape, c["name"]
            assert x.is_cuda and y.is_cuda
            np.testing.assert_array_equal(y.cpu().numpy(), gy, err_msg=c["name"])
            np.testing.assert_allclose(x.cpu().numpy(), gx, rtol=0, atol=2e-5, err_msg=c["name"])


# ------------------------------------------------------------------------------------------------ gaussian filter, shot noise (skimage family)
def test_gaussian_filter_augmentation_vs_the_reference_outputs(ops, golden_dir):
    """fmri_correlate1d_f32 (mode 'nearest', fp64 sums, fp32 storage) x 3 axes against what the reference's apply_gaussian_filter returned
    over scikit-image 0.18.3 (tests/golden/skimage_golden.npz): one fp32 rounding of input and output"""
    z = np.load(os.path.join(golden_dir, "skimage_golden.npz"))
    n = 0
    while "gfilter_%d" % n in z.files:
        vol, sigma, want = z["in_" + str(z["gfilter_%d_in" % n])], float(z["gfilter_%d_args" % n][0]), z["gfilter_%d" % n]
        t = torch.from_numpy(vol.astype(np.float32)).cuda()
        got = ops.gaussian_filter_f32(t, sigma).cpu().numpy().astype(np.float64)
        scale = max(np.abs(want).max(), 1e-30)
        assert np.abs(got - want).max() <= 3e-7 * scale, (n, np.abs(got - want).max() / scale)
        n += 1
    assert n == 5
    # a larger patch, a radius beyond the extent of an axis (sigma 2.5 -> radius 10 on an 8-long axis), the RGB rule of skimage
    rs = np.random.RandomState(4)
    for shape, sigma in (((40, 24, 8), 2.5), ((16, 16, 3), 1.2), ((9, 7, 5), 0.4)):
        vol = rs.randn(*shape).astype(np.float32)
        got = ops.gaussian_filter_f32(torch.from_numpy(vol).cuda(), sigma).cpu().numpy()
        want = OA.apply_gaussian_filter(vol.astype(np.float64), sigma)
        assert np.abs(got - want).max() <= 3e-7 * np.abs(want).max(), shape


def test_shot_noise_vs_oracle_with_the_same_poisson_draws(ops):
    """the deterministic frame of the reference's shot_noise (min-max scaling, 1023-level quantisation, skimage's power-of-two `vals`,
    clip, inverse scaling) with the device's rates handed to numpy's Poisson generator - the oracle consumes the very same draws"""
    rs = np.random.RandomState(9)
    for shape, gen in (((12, 10, 6), lambda: rs.randn(12, 10, 6) * 40 + 100), ((8, 8, 4), lambda: np.round(rs.rand(8, 8, 4) * 5) / 5)):
        x = gen().astype(np.float32)
        t = torch.from_numpy(x.copy()).cuda()
        stats = torch.empty(2, device="cuda")
        ops.minmax(t, stats)
        seen = {}

        def draws_fn(rates):
            lam = rates.cpu().numpy().astype(np.float64)
            seen["lam"] = lam
            seen["draw"] = np.random.RandomState(21).poisson(lam)
            return torch.from_numpy(seen["draw"].astype(np.float32)).cuda()

        ops.shot_noise(t, stats, draws_fn=draws_fn)
        got = t.cpu().numpy().astype(np.float64)
        # oracle on the fp32 values, fed the same draws; its rates must agree with the device's except where fp32 rounding moves a
        # value across a quantisation boundary (none in these volumes)
        want_rates = {}
        want = OA.shot_noise(x.astype(np.float64), poisson=lambda lam: (want_rates.setdefault("lam", lam), seen["draw"].reshape(lam.shape))[1])
        lam_dev, lam_ref = seen["lam"].reshape(x.shape), want_rates["lam"]
        off = np.abs(lam_dev - lam_ref) > 1e-3 * max(lam_ref.max(), 1.0)
        assert off.mean() <= 2e-3, off.mean()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-6 * scale
        # skimage's rule for the number of levels: a power of two >= the number of distinct quantised values
        levels = len(np.unique(np.floor(np.clip((x.astype(np.float64) - x.min()) / (x.max() - x.min()), 0, 1) * 1023)))
        vals = 2 ** int(np.ceil(np.log2(levels)))
        assert abs(lam_dev.max() - vals) <= 1e-3 * vals                 # the maximum (scaled value 1.0) has rate exactly vals


def test_shot_noise_device_draws_are_poisson(ops):
    """with torch's device generator: mean and variance of the scaled draws match the Poisson law (statistical check, fixed seed)"""
    x = torch.linspace(0.0, 1.0, 1024, device="cuda").repeat(256).contiguous()          # all 1024 levels occupied -> vals = 1024
    stats = torch.empty(2, device="cuda")
    ops.minmax(x, stats)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    ref = x.clone()
    rates = ops.shot_noise(x, stats, generator=g)
    torch.cuda.synchronize()
    assert abs(float(rates.max()) - 1024.0) < 1e-2
    sel = (ref > 0.45) & (ref < 0.55)
    lam = rates[sel].double()
    d = (x[sel] * 1024.0).double()                          # the draws themselves (the band is far from the clip at 1.0)
    assert abs(float((d - lam).mean())) < 0.01 * float(lam.mean())
    assert abs(float(((d - lam) ** 2).mean()) / float(lam.mean()) - 1.0) < 0.05                   # Poisson: variance == mean
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0


def test_device_generator_applies_gaussian_filter_and_poisson_noise():
    """the two augmenters in the generator: always-on configuration smooths the patch (lower high-frequency energy than the plain patch
    of the same seed) and the labels are untouched; no 'not applied' warning is raised for them any more"""
    import warnings
    from fetal_net.device_generator import device_data_generator
    vols, truths = synth_volumes(3, [(40, 40, 24)])
    df = FakeDataFile(vols, truths)
    outs = {}
    for tag, aug in (("plain", {"flip": [0, 0, 0]}), ("aug", {"flip": [0, 0, 0], "gaussian_filter": {"prob": 1.0, "max_sigma": 1.5}, "poisson_noise": 1.0})):
        np.random.seed(2)
        random.seed(2)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            gen = device_data_generator(df, [0], batch_size=1, augment=aug, patch_shape=(16, 16, 8), skip_blank=False, categorical=False,
                                        is3d=True, truth_index=0, truth_size=8, shuffle_index_list=False)
            x, y = next(gen)
        outs[tag] = (x.float().cpu().numpy(), y.cpu().numpy())
    (xp, yp), (xa, ya) = outs["plain"], outs["aug"]
    assert xp.shape == xa.shape and np.isfinite(xa).all()
    rough = lambda a: np.abs(np.diff(a, axis=2)).mean()
    assert not np.allclose(xp, xa)
    assert xa.min() >= xp.min() - 1e-3 * abs(xp.min()) - 1e-3 and xa.max() <= xp.max() + 1e-3 * abs(xp.max()) + 1e-3      # smoothing + clipped noise stay in range


# ---------------------------------------------------------------------------------------------- imgaug's augmenters (oracle: parity unpinned)
@pytest.mark.parametrize("seed,shape,sigma,alpha", [(0, (64, 48, 5), 10.0, 5.0), (1, (33, 70, 3), 2.0, 40.0), (2, (40, 40, 1), 4.0, 120.0)])
def test_elastic_fields_and_warp_vs_oracle(ops, seed, shape, sigma, alpha):
    """same uniform noise -> same displacement fields (fp32 correlation on the device, fp64 in the oracle); same fields -> the warped image
    (bilinear) to 1e-6 of its range and the warped labels (nearest) bit for bit, also into a strided destination slot"""
    from oracle import augment_oracle as AO
    rs = np.random.RandomState(seed)
    X, Y, C = shape
    k = AO.elastic_ksize(sigma)
    assert k == ops.elastic_ksize(sigma) and k % 2 == 1
    noise = (rs.rand(2, X + 2 * k, Y + 2 * k) * 2 - 1).astype(np.float32)
    dx, dy = AO.elastic_shift_maps((X, Y), alpha, sigma, noise)
    d0, d1 = ops.elastic_fields((X, Y), alpha, sigma, noise=torch.from_numpy(noise).cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(d0.cpu().numpy(), dy, rtol=0, atol=2e-6 * alpha)
    np.testing.assert_allclose(d1.cpu().numpy(), dx, rtol=0, atol=2e-6 * alpha)
    assert np.abs(dx).max() > 0.05                                     # a field that moves something
    # warp with the DEVICE's fields on both sides (the oracle takes them as float64)
    f0, f1 = d0.cpu().numpy().astype(np.float64), d1.cpu().numpy().astype(np.float64)
    img = rs.randn(X, Y, C).astype(np.float32)
    lab = (rs.rand(X, Y, C) > 0.6).astype(np.uint8)
    want_img, want_lab = AO.elastic_apply(img, f1, f0, 1), AO.elastic_apply(lab, f1, f0, 0)
    wide = torch.full((X, Y, C + 2), -7.0, device="cuda")              # destination = the first C channels of a wider row
    ops.elastic_warp(torch.from_numpy(img).cuda(), d0, d1, 1, wide[..., :C])
    got_lab = ops.elastic_warp(torch.from_numpy(lab).cuda(), d0, d1, 0, torch.empty((X, Y, C), device="cuda", dtype=torch.uint8))
    torch.cuda.synchronize()
    w = wide.cpu().numpy()
    np.testing.assert_allclose(w[..., :C], want_img, rtol=0, atol=2e-6 * float(np.abs(img).max()))
    assert (w[..., C:] == -7.0).all()
    # nearest: a coordinate within 1e-6 of a half-integer may round either way in fp32-derived fields; none do here
    assert np.array_equal(got_lab.cpu().numpy(), want_lab.astype(np.uint8))


def test_elastic_zero_field_is_the_identity_and_borders_clamp(ops):
    X, Y, C = 12, 9, 2
    img = torch.randn(X, Y, C, device="cuda")
    z = torch.zeros(X, Y, device="cuda")
    out = ops.elastic_warp(img, z, z, 1, torch.empty_like(img))
    assert torch.equal(out, img)
    far = torch.full((X, Y), 100.0, device="cuda")                     # every voxel reads from beyond the top-left corner: clamped to it
    out = ops.elastic_warp(img, far, far, 1, torch.empty_like(img))
    assert torch.equal(out, img[0:1, 0:1, :].expand(X, Y, C))


@pytest.mark.parametrize("per_channel", [True, False])
def test_coarse_dropout_vs_oracle(ops, per_channel):
    from oracle import augment_oracle as AO
    rs = np.random.RandomState(3)
    X, Y, C = 50, 37, 6
    img = (rs.randn(X, Y, C) * 3 + 1).astype(np.float32)
    hs, ws = AO.coarse_dropout_grid((X, Y), [0.10, 0.30], rs)
    assert hs in (5, 15) and ws in (3, 11)
    keep = (rs.rand(hs, ws, C if per_channel else 1) >= 0.2).astype(np.uint8)
    want = AO.coarse_dropout(img, keep)
    x = torch.from_numpy(img).cuda()
    stats = torch.empty(2, device="cuda")
    ops.minmax(x, stats)
    ops.coarse_dropout(x, torch.from_numpy(keep).cuda(), stats, per_channel)
    torch.cuda.synchronize()
    got = x.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6 * float(np.abs(img).max()))
    dropped = got == img.min()
    assert 0.05 < dropped.mean() < 0.45 and np.array_equal(got[~dropped], img[~dropped])       # kept voxels are untouched, bit for bit


@pytest.mark.parametrize("masks", [False, True])
def test_device_generator_batched_launches_and_prefetch_thread_yield_the_same_batches(masks):
    """three ways to the same batches, bit for bit: patch by patch (batched=False: the form every other test of this file checks against the
    oracle and the reference's fixtures), the patches of a batch launched together (the default), and that with a producer thread on a stream of
    its own keeping two batches ready (prefetch=2) while the consumer reads on alternating streams.  The reference's default augmentation is on;
    the second volume's labels are almost empty, so skip_blank's read-back drops patches and the kept ones move up.  (shuffle off: the shuffling
    index generator re-seeds numpy from the OS on every pass, as the reference's does.)"""
    import threading
    from fetal_net.device_generator import device_data_generator
    default = {"flip": [0.5, 0.5, 0.5], "permute": False, "translate": (15, 15, 7), "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "poisson_noise": 1,
               "gaussian_filter": {"prob": 0.0, "max_sigma": 1}, "contrast": {"prob": 0, "min_factor": 0.2, "max_factor": 0.1},
               "elastic_transform": {"alpha": 5, "sigma": 10},
               "coarse_dropout": {"rate": 0.2, "size_percent": [0.10, 0.30], "per_channel": True},
               "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05}}
    vols, truths = synth_volumes(3, [(72, 72, 40), (64, 80, 36)])
    truths[1][:] = 0
    truths[1][20:30, 30:44, 10:20] = 1                                  # most patches of volume 1 are blank
    mk = [np.random.RandomState(4).rand(*t.shape).astype(np.float32) for t in truths] if masks else None
    df = FakeDataFile(vols, truths, mk)
    runs, dropped = [], 0
    for batched, prefetch in ((False, 0), (True, 0), (True, 2)):
        np.random.seed(11)
        random.seed(11)
        gen = device_data_generator(df, [0, 1], batch_size=3, augment=default, patch_shape=(48, 48, 16), skip_blank=True, categorical=True, is3d=True,
                                    truth_index=0, truth_size=16, noise_seed=3, prefetch=prefetch, batched=batched, shuffle_index_list=False)
        got = []
        reader = torch.cuda.Stream()
        for k in range(4):
            with torch.cuda.stream(reader if k % 2 else torch.cuda.current_stream()):
                x, y = next(gen)
                got.append([t.clone() for t in (x if masks else [x])] + [y.clone()])
        torch.cuda.synchronize()
        gen.close()                                                      # stops and joins the producer thread
        assert not [t for t in threading.enumerate() if t.name == "device_data_generator"]
        runs.append([[t.cpu().numpy() for t in b] for b in got])
    for variant in runs[1:]:
        for b0, b1 in zip(runs[0], variant):
            assert len(b0) == len(b1) == (3 if masks else 2)
            for t0, t1 in zip(b0, b1):
                assert t0.shape == t1.shape and np.array_equal(t0, t1)
    assert not np.array_equal(runs[0][0][0], runs[0][1][0])
    assert all(b[-1].reshape(3, -1, 2)[..., 1].any(axis=1).all() for b in runs[0])      # no blank patch was kept


def test_device_generator_batches_beyond_one_launch_chunk_equal_patch_by_patch():
    """19 patches per batch: the batch kernels take 16 patches per launch (per-patch parameters travel by value), so this batch is two
    chunks of every step - pointers, statistics and workspaces of the second chunk offset by hand in the C layer"""
    from fetal_net.device_generator import device_data_generator
    default = {"flip": [0.5, 0.5, 0.5], "translate": (5, 5, 3), "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "poisson_noise": 1,
               "contrast": {"prob": 0, "min_factor": 0.2, "max_factor": 0.1}, "elastic_transform": {"alpha": 5, "sigma": 4},
               "coarse_dropout": {"rate": 0.2, "size_percent": [0.10, 0.30], "per_channel": True},
               "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05}}
    vols, truths = synth_volumes(9, [(40, 44, 24), (36, 50, 30), (48, 40, 20)])
    mk = [np.random.RandomState(5).rand(*t.shape).astype(np.float32) for t in truths]
    df = FakeDataFile(vols, truths, mk)
    outs = []
    for batched in (False, True):
        np.random.seed(21)
        random.seed(21)
        gen = device_data_generator(df, [0, 1, 2], batch_size=19, augment=default, patch_shape=(32, 32, 8), skip_blank=False, categorical=False,
                                    is3d=True, truth_index=0, truth_size=8, batched=batched, shuffle_index_list=False, noise_seed=4)
        b = []
        for _ in range(2):
            (x, m), y = next(gen)
            b += [x.cpu().numpy(), m.cpu().numpy(), y.cpu().numpy()]
        outs.append(b)
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert not np.array_equal(outs[0][0][16], outs[0][0][3])


def test_device_generator_without_augmentation_batched_equals_patch_by_patch():
    from fetal_net.device_generator import device_data_generator
    vols, truths = synth_volumes(5, [(40, 44, 24), (36, 50, 30)])
    mk = [np.random.RandomState(4).rand(*t.shape).astype(np.float32) for t in truths]
    df = FakeDataFile(vols, truths, mk)
    outs = []
    for batched in (False, True):
        np.random.seed(2)
        random.seed(2)
        gen = device_data_generator(df, [0, 1], batch_size=5, augment=None, patch_shape=(32, 32, 8), skip_blank=False, categorical=False, is3d=False,
                                    truth_index=3, truth_size=2, batched=batched, shuffle_index_list=False)
        (x, m), y = next(gen)
        outs.append([t.cpu().numpy() for t in (x, m, y)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_device_generator_runs_the_reference_default_augmentation_without_warnings():
    """fetal/config_utils.py:81-123 verbatim: elastic transform and coarse dropout included - no 'not applied' warning; labels stay binary,
    the image stays inside the volume's range, about `rate` of the voxels sit at the patch minimum"""
    import warnings
    from fetal_net.device_generator import device_data_generator
    default = {"flip": [0.5, 0.5, 0.5], "permute": False, "translate": (15, 15, 7), "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "poisson_noise": 1,
               "gaussian_filter": {"prob": 0.0, "max_sigma": 1}, "contrast": {"prob": 0, "min_factor": 0.2, "max_factor": 0.1},
               "elastic_transform": {"alpha": 5, "sigma": 10},
               "coarse_dropout": {"rate": 0.2, "size_percent": [0.10, 0.30], "per_channel": True},
               "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05}}
    vols, truths = synth_volumes(3, [(72, 72, 40)])
    df = FakeDataFile(vols, truths)
    np.random.seed(5)
    random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        gen = device_data_generator(df, [0], batch_size=2, augment=default, patch_shape=(48, 48, 16), skip_blank=False, categorical=False, is3d=True,
                                    truth_index=0, truth_size=16, shuffle_index_list=False)
        x, y = next(gen)
        x2, _ = next(gen)
    x, y = x.float().cpu().numpy(), y.cpu().numpy()
    assert x.shape == (2, 1, 48, 48, 16) and np.isfinite(x).all() and set(np.unique(y)) <= {0, 1}
    assert not np.array_equal(x, x2.float().cpu().numpy())
    # the dropped share, measured without the geometric part (a rotated patch is also at the volume's minimum wherever it leaves the volume)
    still = dict(default, rotate=None, translate=None, scale=None, flip=[0, 0, 0])
    np.random.seed(6)
    random.seed(6)
    xs, _ = next(device_data_generator(df, [0], batch_size=4, augment=still, patch_shape=(48, 48, 16), skip_blank=False, categorical=False, is3d=True,
                                       truth_index=0, truth_size=16, shuffle_index_list=False))
    xs = xs.float().cpu().numpy()
    at_min = float(np.mean([(xs[b] == xs[b].min()).mean() for b in range(4)]))
    assert 0.10 < at_min < 0.32, at_min                               # Binomial(0.8) keeps on 4 / 14-cell grids, one per slice: 0.2 on average
    with warnings.catch_warnings():                                   # the commented-out entry of the default config, switched on: applied too
        warnings.simplefilter("error")
        xp_, yp_ = next(device_data_generator(df, [0], batch_size=1, augment=dict(default, piecewise_affine={"scale": 2}), patch_shape=(48, 48, 16),
                                              skip_blank=False, categorical=False, is3d=True, truth_index=0, truth_size=16, shuffle_index_list=False))
    assert np.isfinite(xp_.float().cpu().numpy()).all() and set(np.unique(yp_.cpu().numpy())) <= {0, 1}


def test_elastic_moves_image_and_labels_together():
    """a strong field (alpha 300, sigma 6) through the generator: the label patch still marks the bright blob of the image patch - the
    same field warps both (reference augment.py:149-170: four augmenters over one random state)"""
    from fetal_net.device_generator import device_data_generator
    rs = np.random.RandomState(0)
    X = 64
    g = np.stack(np.meshgrid(*[np.arange(X)] * 3, indexing="ij"))
    blob = (((g[:2] - X / 2) ** 2).sum(0) < (X / 4) ** 2)           # a cylinder along z: every slice of any patch crosses its boundary
    vol = blob * 4.0 + rs.randn(X, X, X) * 0.05
    df = FakeDataFile([vol.astype(np.float32)], [blob.astype(np.uint8)])
    outs = {}
    for tag, aug in (("plain", {"flip": [0, 0, 0]}), ("elastic", {"flip": [0, 0, 0], "elastic_transform": {"alpha": 600, "sigma": 6}})):
        np.random.seed(1)
        random.seed(1)
        gen = device_data_generator(df, [0], batch_size=1, augment=aug, patch_shape=(48, 48, 8), skip_blank=False, categorical=False, is3d=True,
                                    truth_index=0, truth_size=8, shuffle_index_list=False, noise_seed=4)
        x, y = next(gen)
        outs[tag] = (x.float().cpu().numpy()[0, 0], y.cpu().numpy()[0, 0])
    (xp, yp), (xe, ye) = outs["plain"], outs["elastic"]
    assert (yp != ye).mean() > 0.005                                   # the field moved the boundary
    agree = ((xe > 2.0) == (ye > 0)).mean()
    assert agree > 0.985, agree                                        # bilinear image vs nearest label: they differ on the boundary voxels only


def test_elastic_warps_mask_and_previous_slice_truth_with_the_same_field():
    """2-D flow with the previous-slice truth in the input channels and distance masks in the data file (reference generator.py:18-21,
    :300-328): under a strong elastic field the previous-slice truth channel (a label slice) and the mask stay consistent with the warped
    labels: with truth_index = prev_truth_index the extra input channel IS the label patch, and the mask of a binary 0 / 1 'mask volume'
    equal to the labels IS the label patch as float"""
    from fetal_net.device_generator import device_data_generator
    X = 64
    g = np.stack(np.meshgrid(*[np.arange(X)] * 3, indexing="ij"))
    blob = (((g[:2] - X / 2) ** 2).sum(0) < (X / 4) ** 2)
    vol = (blob * 4.0 + np.random.RandomState(0).randn(X, X, X) * 0.05).astype(np.float32)
    lab = blob.astype(np.uint8)
    pad = 3                                                             # samples_pad: labels are padded with the volumes, masks are not
    mask = np.pad(lab.astype(np.float32), ((pad, pad), (pad, pad), (pad, pad)))
    df = FakeDataFile([vol], [lab], [mask])
    np.random.seed(3)
    random.seed(3)
    gen = device_data_generator(df, [0], batch_size=2, augment={"flip": [0, 0, 0], "elastic_transform": {"alpha": 600, "sigma": 6}},
                                patch_shape=(48, 48, 5), skip_blank=False, categorical=False, is3d=False, truth_index=2, truth_size=1,
                                prev_truth_index=2, prev_truth_size=1, shuffle_index_list=False, noise_seed=9)
    (x, m), y = next(gen)
    torch.cuda.synchronize()
    x, m, y = x.cpu().numpy(), m.cpu().numpy(), y.cpu().numpy()
    assert x.shape == (2, 48, 48, 6) and m.shape == (2, 48, 48, 1) and y.shape == (2, 48, 48, 1)
    assert 0.02 < y.mean() < 0.98                                       # the patches cross the cylinder's boundary
    assert np.array_equal(x[..., 5:6], y.astype(np.float32))             # previous-slice truth channel = the (warped) label slice
    assert np.array_equal(m, y.astype(np.float32))                       # mask volume = labels: warped by the same field, same rounding
    plain = device_data_generator(df, [0], batch_size=2, augment={"flip": [0, 0, 0]}, patch_shape=(48, 48, 5), skip_blank=False, categorical=False,
                                  is3d=False, truth_index=2, truth_size=1, prev_truth_index=2, prev_truth_size=1, shuffle_index_list=False)
    np.random.seed(3)
    random.seed(3)
    (_, _), y0 = next(plain)
    assert (y0.cpu().numpy() != y).mean() > 0.003                        # and the field did move the labels


@pytest.mark.parametrize("seed,shape,scale", [(0, (48, 64, 3), 0.05), (1, (64, 48, 2), 0.2), (2, (40, 40, 1), 2.0)])
def test_piecewise_affine_vs_oracle(ops, seed, shape, scale):
    """the two-triangle warp of the device (closed-form triangle maps, diagonal test) against the oracle's generic route (scipy Delaunay of the
    source grid, one solved affine per simplex, map_coordinates): image to 1e-5 of its range - a coordinate within 1e-12 of the image border or of a
    half-integer may fall either way, so a handful of voxels are allowed to differ - labels likewise"""
    from oracle import augment_oracle as AO
    rs = np.random.RandomState(seed)
    X, Y, C = shape
    src, dst = AO.piecewise_affine_points((X, Y), rs.normal(0, scale, size=(4, 2)))
    assert np.array_equal(src, [[0, 0], [0, Y], [X, 0], [X, Y]]) and dst[:, 0].max() <= X - 1 and dst[:, 1].max() <= Y - 1 and dst.min() >= 0
    img = (rs.rand(X, Y, C) + 0.5).astype(np.float32)                  # strictly positive: a voxel read from outside (cval 0) is recognisable
    lab = (rs.rand(X, Y, C) > 0.5).astype(np.uint8)
    want_img, want_lab = AO.piecewise_affine_apply(img, src, dst, 1), AO.piecewise_affine_apply(lab, src, dst, 0)
    got_img = ops.piecewise_affine(torch.from_numpy(img).cuda(), dst, 1, torch.empty((X, Y, C), device="cuda"))
    got_lab = ops.piecewise_affine(torch.from_numpy(lab).cuda(), dst, 0, torch.empty((X, Y, C), device="cuda", dtype=torch.uint8))
    torch.cuda.synchronize()
    bad_img = np.abs(got_img.cpu().numpy() - want_img) > 1e-5 * 1.5
    bad_lab = got_lab.cpu().numpy() != want_lab.astype(np.uint8)
    assert bad_img.mean() <= 2e-3 and bad_lab.mean() <= 2e-3, (bad_img.mean(), bad_lab.mean())
    assert (want_img != 0).mean() > 0.2                                # the warp keeps a good part of the image in view


def test_piecewise_affine_with_unmoved_corners_is_a_crop_of_the_identity(ops):
    X, Y, C = 20, 30, 2
    img = torch.rand(X, Y, C, device="cuda") + 1.0
    out = ops.piecewise_affine(img, [[0, 0], [0, Y], [X, 0], [X, Y]], 1, torch.empty_like(img))
    assert torch.allclose(out, img, atol=1e-6)
    # corners pulled to the image's own corner voxels (what the clip to [0, h-1] x [0, w-1] does to an unmoved grid): a slight zoom, nothing from outside
    out = ops.piecewise_affine(img, [[0, 0], [0, Y - 1], [X - 1, 0], [X - 1, Y - 1]], 1, torch.empty_like(img))
    assert float(out.min()) >= 1.0 - 1e-6


# ---------------------------------------------------------------------------------------------- in-kernel draws + chained min / max (fmri_*_rng)
def _ws_is_armed(ws):
    return not bool(ws.any().item())


@pytest.mark.parametrize("n,dtype,offset", [(1, torch.float32, 0), (1000003, torch.float32, 0), (4096, torch.bfloat16, 0), (12345, torch.bfloat16, 1),
                                            (70001, torch.float32, 3)])
def test_minmax_ws_one_launch_equals_torch_and_rearms_its_workspace(ops, n, dtype, offset):
    stats, ws = ops.aug_workspace("cuda")
    base = (torch.randn(n + offset, device="cuda") * 3 + 1).to(dtype)
    x = base[offset:]                                               # offset > 0: a base address that is not 16-byte aligned
    for _ in range(3):                                              # the same workspace, call after call
        ops.minmax_ws(x, stats, ws)
        assert stats.tolist() == [float(x.float().min()), float(x.float().max())]
        assert _ws_is_armed(ws)
        x = (x.float() * -0.5 + 2).to(dtype)


def _moments(e):
    e = e.astype(np.float64)
    m, v = e.mean(), e.var()
    return m, v, ((e - m) ** 3).mean() / v ** 1.5, ((e - m) ** 4).mean() / v ** 2


@pytest.mark.parametrize("kind", [0, 1])
def test_noise_rng_draws_are_standard_normal_and_the_range_is_chained(ops, kind):
    """x' = clip(s + e) (gaussian) or clip(s + s e) (speckle) with e = sigma N(0,1), on an image scaled to [0, 1] by its own min / max (exactly 0
    and 1 are in it): where nothing clips, (x' - x) / sigma [/ x] must be standard normal - moments, a chi-square over 40 bins, no correlation
    between neighbours - the same for the same (seed, seq), different for another seq; `stats` afterwards = the new image's min / max"""
    n, sigma = 1 << 20, 0.05
    rs = np.random.RandomState(7)
    x0 = (rs.rand(n) * 0.4 + 0.3).astype(np.float32)
    x0[5], x0[77] = 0.0, 1.0
    stats, ws = ops.aug_workspace("cuda")
    outs = []
    for seq in (1, 1, 2):
        x = torch.from_numpy(x0).cuda()
        ops.minmax_ws(x, stats, ws)
        ops.noise_rng(x, stats, ws, kind, sigma, 1234, seq)
        assert stats.tolist() == [float(x.min()), float(x.max())] and _ws_is_armed(ws)
        outs.append(x.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
    keep = np.ones(n, bool)
    keep[[5, 77]] = False
    e = (outs[0] - x0)[keep].astype(np.float64) / sigma
    if kind == 1:
        e = e / x0[keep]
    m, v, sk, ku = _moments(e)
    assert abs(m) < 5e-3 and abs(v - 1) < 1e-2 and abs(sk) < 2e-2 and abs(ku - 3) < 5e-2, (m, v, sk, ku)
    assert abs(np.corrcoef(e[:-1], e[1:])[0, 1]) < 5e-3 and abs(np.corrcoef(e[:-256], e[256:])[0, 1]) < 5e-3
    import scipy.stats
    edges = scipy.stats.norm.ppf(np.linspace(0, 1, 41)[1:-1])
    counts = np.bincount(np.searchsorted(edges, e), minlength=40)
    chi2 = ((counts - e.size / 40.0) ** 2 / (e.size / 40.0)).sum()
    assert chi2 < scipy.stats.chi2.ppf(1 - 1e-6, 39), chi2          # fp32 rounding of x' - x is ~1e-6 / sigma of a bin edge: invisible at this n
    # bf16 image: runs, stays in range, chained range is that of the stored (rounded) values
    xb = torch.from_numpy(x0).cuda().to(torch.bfloat16)
    ops.minmax_ws(xb, stats, ws)
    ops.noise_rng(xb, stats, ws, kind, sigma, 1234, 3)
    assert stats.tolist() == [float(xb.float().min()), float(xb.float().max())] and 0.0 <= stats[0].item() and stats[1].item() <= 1.0


def test_shot_noise_rng_draws_are_poisson_on_both_branches_of_the_sampler(ops):
    """reference augment.py:87-94 with the draw made in the kernel: the image is quantised to its 1,024 levels, vals = 2 ** ceil(log2(levels in use)),
    x' = Poisson(level / 1023 * vals) / vals clipped to [0, 1].  Recover the integer draws and test them per rate against the Poisson law:
    mean and variance for every rate, the full histogram (chi-square) for four rates either side of the sampler's switch at lam = 10."""
    import scipy.stats
    n = 1 << 21
    rs = np.random.RandomState(3)
    lev = rs.randint(0, 600, size=n)
    lev[0], lev[1] = 0, 1023
    x0 = ((lev + 0.5) / 1023.0).astype(np.float32)
    x0[0], x0[1] = 0.0, 1.0
    level = np.floor(np.clip(x0, 0, 1) * np.float32(1023.0)).astype(np.int64)
    assert np.array_equal(level, lev)
    vals = 1024                                                      # 601 levels in use
    stats, ws = ops.aug_workspace("cuda")
    outs = []
    for seq in (9, 9, 10):
        x = torch.from_numpy(x0).cuda()
        ops.minmax_ws(x, stats, ws)
        ops.shot_noise_rng(x, stats, ws, 99, seq)
        assert stats.tolist() == [float(x.min()), float(x.max())] and _ws_is_armed(ws)
        outs.append(x.cpu().numpy().astype(np.float64))
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
    k = outs[0] * vals
    assert np.abs(k - np.round(k)).max() < 1e-3
    k = np.round(k).astype(np.int64)
    lam = level / 1023.0 * vals
    for L in range(0, 600, 7):
        sel = k[level == L]
        lm = L / 1023.0 * vals
        se = np.sqrt(max(lm, 1e-9) / sel.size)
        assert abs(sel.mean() - lm) < 5 * se + 1e-9, (L, sel.mean(), lm)
        if L:
            assert abs(sel.var() / lm - 1) < 6 * np.sqrt(2.0 / sel.size) + 6 / np.sqrt(lm * sel.size), (L, sel.var(), lm)
    assert (k[level == 0] == 0).all()
    for L in (3, 9, 10, 11, 40, 599):                               # lam = 3.0, 9.0 (products of uniforms), 10.01, 11.0, 40, 599.6 (PTRS)
        sel = np.concatenate([k[level == L]] + [np.round(o * vals).astype(np.int64)[level == L] for o in outs[2:]])
        lm = L / 1023.0 * vals
        lo_, hi_ = int(scipy.stats.poisson.ppf(1e-3, lm)), int(scipy.stats.poisson.ppf(1 - 1e-3, lm))
        edges = np.arange(lo_, hi_ + 1)
        pm = scipy.stats.poisson.pmf(edges, lm)
        pm = np.concatenate([[scipy.stats.poisson.cdf(lo_ - 1, lm)], pm, [scipy.stats.poisson.sf(hi_, lm)]])
        cnt = np.bincount(np.clip(sel, lo_ - 1, hi_ + 1) - (lo_ - 1), minlength=pm.size)
        ok = pm * sel.size > 5
        chi2 = ((cnt[ok] - pm[ok] * sel.size) ** 2 / (pm[ok] * sel.size)).sum()
        assert chi2 < scipy.stats.chi2.ppf(1 - 1e-5, ok.sum() - 1), (L, chi2, ok.sum())
    # a second image through the SAME workspace with fewer levels in use: the level table was cleared behind the first
    x = torch.from_numpy(((rs.randint(0, 5, size=4096) * 255.75) / 1023.0).astype(np.float32)).cuda()
    ops.minmax_ws(x, stats, ws)
    ops.shot_noise_rng(x, stats, ws, 99, 11)
    kk = x.cpu().numpy().astype(np.float64) * 8                      # 5 levels -> vals = 8
    assert np.abs(kk - np.round(kk)).max() < 1e-4 and _ws_is_armed(ws)


@pytest.mark.parametrize("per_channel", [True, False])
def test_coarse_dropout_rng_drops_whole_cells_at_the_given_rate(ops, per_channel):
    X, Y, C, hs, wsz, rate = 64, 128, 128, 7, 13, 0.2
    x0 = torch.randn(X, Y, C, device="cuda")
    stats, ws = ops.aug_workspace("cuda")
    ops.minmax_ws(x0, stats, ws)
    x = ops.coarse_dropout_rng(x0.clone(), (hs, wsz), rate, stats, per_channel, 5, 1)
    again = ops.coarse_dropout_rng(x0.clone(), (hs, wsz), rate, stats, per_channel, 5, 1)
    other = ops.coarse_dropout_rng(x0.clone(), (hs, wsz), rate, stats, per_channel, 5, 2)
    assert torch.equal(x, again) and not torch.equal(x, other)
    got, src = x.cpu().numpy(), x0.cpu().numpy()
    dropped = got != src
    assert (got[dropped] == src.min()).all()
    si = np.minimum(np.floor(np.arange(X) * (hs / X)).astype(int), hs - 1)
    sj = np.minimum(np.floor(np.arange(Y) * (wsz / Y)).astype(int), wsz - 1)
    cells = np.zeros((hs, wsz, C), bool)
    seen = np.zeros((hs, wsz, C), bool)
    for i in range(X):
        for j in range(Y):
            d = dropped[i, j] | (src[i, j] == src.min())
            if seen[si[i], sj[j], 0]:
                same = (cells[si[i], sj[j]] == dropped[i, j]) | (src[i, j] == src.min())
                assert same.all()                                     # one decision per cell (and slice)
            else:
                cells[si[i], sj[j]] = dropped[i, j]
                seen[si[i], sj[j]] = True
            del d
    if not per_channel:
        assert (cells == cells[:, :, :1]).all()
    nd = cells[:, :, 0].size if not per_channel else cells.size
    frac = (cells[:, :, 0] if not per_channel else cells).mean()
    assert abs(frac - rate) < 4 * np.sqrt(rate * (1 - rate) / nd) + 1e-9, frac


def test_rescale_intensity_ws_equals_the_plain_pass_and_chains_the_range(ops):
    x0 = torch.randn(70001, device="cuda") * 2 + 1
    stats, ws = ops.aug_workspace("cuda")
    for contrast, lo, hi, mult in ((True, -1.0, 2.5, 1.0), (False, 0.0, 0.0, 1.2), (True, 0.5, 0.5, 0.9)):
        a, b = x0.clone(), x0.clone()
        ops.minmax_ws(a, stats, ws)
        before = stats.clone()
        ops.rescale_intensity(a, before, contrast, lo, hi, mult)
        ops.rescale_intensity_ws(b, stats, ws, contrast, lo, hi, mult)
        assert torch.equal(a, b) and stats.tolist() == [float(b.min()), float(b.max())] and _ws_is_armed(ws)


@pytest.mark.parametrize("X,Y,sigma,alpha", [(64, 128, 10.0, 5.0), (33, 150, 4.0, 40.0), (9, 7, 2.0, 1.5)])
def test_elastic_fields_rng_equals_the_host_restatement_of_its_draws(ops, X, Y, sigma, alpha):
    """the one-launch displacement fields against numpy: the SAME noise (Philox4x32-10 restated in tests/philox_ref.py, pinned to Random123's
    known answers: pixel p of the padded (2, X + 2k, Y + 2k) grid = word p & 3 of block p >> 2), the same truncated Gaussian along both axes
    in fp64, times alpha, cropped.  Also pins the kernels' Philox itself, which the noise kernels share."""
    from philox_ref import philox4x32_10, u01
    seed, seq = 0x1234567890abcdef, 77
    k = ops.elastic_ksize(sigma)
    hp, wp = X + 2 * k, Y + 2 * k
    pix = np.arange(2 * hp * wp, dtype=np.uint64)
    words = np.stack(philox4x32_10(pix >> np.uint64(2), 0, seq, 1, seed & 0xffffffff, seed >> 32), axis=-1)
    noise = (u01(words[np.arange(pix.size), (pix & np.uint64(3)).astype(np.int64)]) * 2 - 1).reshape(2, hp, wp)
    xs = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
    w = np.exp(-(xs * xs) / (2.0 * sigma ** 2))
    w /= w.sum()
    blur = scipy.ndimage.correlate1d(scipy.ndimage.correlate1d(noise, w, axis=1, mode="constant"), w, axis=2, mode="constant")
    want = blur[:, k:k + X, k:k + Y] * alpha
    d0, d1 = ops.elastic_fields_rng((X, Y), alpha, sigma, seed, seq)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d1.cpu().numpy(), want[0], rtol=0, atol=3e-6 * alpha)
    np.testing.assert_allclose(d0.cpu().numpy(), want[1], rtol=0, atol=3e-6 * alpha)
    assert np.abs(want).max() > 0.02 * alpha
    e0, _ = ops.elastic_fields_rng((X, Y), alpha, sigma, seed, seq + 1)
    assert not torch.equal(e0, d0)
