"""CPU restatement (numpy + scipy) of the reference's patch sampler, affine augmentation and test-time augmentation.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).  The product path is fetal_net/device_generator.py +
fetal_net/augment.py + csrc/augment.hip.

Pinned against tests/golden/augment_golden.{npz,json}, which tests/golden/make_augment_fixtures.py produced by running the
reference itself (fetal_net/augment.py, generator.py, utils/utils.py, prediction.py) under stub modules.

Follows:
  reference fetal_net/augment.py:14-84    affine helpers (scale / translate / rotate_x,y,z / flip = 180-degree rotation)
  reference fetal_net/augment.py:189-212  distort_image (centre, flip, scale, rotate, un-centre, translate)
  reference fetal_net/augment.py:222-377  augment_data: order of the random draws and of the operations
  reference fetal_net/utils/utils.py:100-113 interpolate_affine_coords / _range (scipy.ndimage.map_coordinates)
  reference fetal_net/generator.py:13-57,222-328,385-401  DataFileDummy, pad_samples, data_generator, add_data, extract_patch, convert_data
  reference fetal_net/augment.py:380-471  permutation keys, permute_data, reverse_permute_data
  reference fetal_net/prediction.py:19-85,354-367  flip_it, predict_augment, predict_flips, predict, predict_with_permutations
Third-party behaviour restated (libraries absent from the main interpreter): skimage.exposure.rescale_intensity (in_range=(lo,hi),
out_range='image'), skimage.util.random_noise modes 'gaussian' / 'speckle' / 'poisson' (clip to [0,1]), skimage.filters.gaussian, the
sklearn-style MinMaxScaler over the whole array (reference utils.py:116-313).  PINNED: tests/golden/skimage_golden.npz holds outputs of
the reference's own contrast_augment / add_gaussian_noise / add_speckle_noise / shot_noise / apply_gaussian_filter run over
scikit-image 0.18.3 and scikit-learn 0.24.2 (generator tests/golden/make_skimage_fixture.py, run under the container's conda
interpreter); tests/test_oracle_augment.py checks every restatement below against them.
PARITY UNPINNED for the three imgaug augmenters (elastic_*, piecewise_affine_*, coarse_dropout below): the reference's requirements.txt names `imgaug` without a
version and the package is installed nowhere here, so there is neither a fixture nor a run of the reference to pin them to.  They restate
imgaug 0.4.0 (the last release) as published - augmenters/geometric.py ElasticTransformation (`_generate_shift_maps`, `_map_coordinates`) and PiecewiseAffine (`_get_transformer`, over
skimage.transform.PiecewiseAffineTransform / warp),
augmenters/blur.py `blur_gaussian_` / `_compute_gaussian_blur_ksize`, augmenters/arithmetic.py CoarseDropout = MultiplyElementwise over
parameters.FromLowerResolution(Binomial(1 - p)) - anchored on the reference's call sites (fetal_net/augment.py:116-120, :131-170, :373-375).
"""
import itertools
import random

import numpy as np
from scipy import ndimage
from scipy.ndimage import map_coordinates


# ------------------------------------------------------------------------------------------------ affine algebra
def _rot(axis, a):
    s, c = np.sin(a), np.cos(a)
    if axis == 0:
        return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])
    if axis == 1:
        return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


def _translate(A, t):
    A = A.copy()
    A[:3, 3] += np.asarray(t, dtype=np.float64)
    return A


def distort_affine(shape, flip_axis=None, scale_factor=None, rotate_factor=None, translate_factor=None):
    """reference augment.py:189-212 starting from the identity"""
    A = np.eye(4)
    centre = np.array(shape, dtype=np.float64) / 2
    A = _translate(A, -centre)
    if flip_axis is not None:
        for ax in flip_axis:
            A = _rot(int(ax), np.deg2rad(180)).dot(A)
    if scale_factor is not None:
        A = np.diag(list(scale_factor) + [1]).dot(A)
    if rotate_factor is not None:
        for i, a in enumerate(rotate_factor):
            if a != 0:
                A = _rot(i, a).dot(A)
    A = _translate(A, centre)
    if translate_factor is not None:
        A = _translate(A, translate_factor)
    return A


def interpolate_affine_range(data, affine, ranges, order=0, cval=0.0):
    """reference utils.py:100-113: output voxel (i,j,k) of the range samples the source at affine . (i,j,k,1)"""
    grid = np.array(np.meshgrid(*[range(s, e) for s, e in ranges], indexing="ij"), dtype=np.float64)   # (3, nx, ny, nz)
    pts = np.moveaxis(grid, 0, -1) @ affine[:3, :3].T + affine[:3, 3]
    return map_coordinates(data, np.moveaxis(pts, -1, 0), mode="constant", order=order, cval=cval)


# ------------------------------------------------------------------------------------------------ intensity operations
def contrast_augment(data, lo, hi):
    data = np.asarray(data, dtype=np.float64)
    omin, omax = float(data.min()), float(data.max())
    out = np.clip(data, lo, hi)
    if lo != hi:
        return (out - lo) / (hi - lo) * (omax - omin) + omin
    return np.clip(out, omin, omax)


def _minmax01(data):
    dmin, dmax = float(np.min(data)), float(np.max(data))
    rng = dmax - dmin
    scale = 1.0 / (rng if rng != 0 else 1.0)
    mn = 0.0 - dmin * scale
    return np.clip(data * scale + mn, 0.0, 1.0), scale, mn


def add_gaussian_noise(data, sigma, noise):
    """`noise` ~ N(0,1) of data's shape (the reference draws it inside skimage.util.random_noise)"""
    s, scale, mn = _minmax01(np.asarray(data, dtype=np.float64))
    return (np.clip(s + noise * sigma, 0.0, 1.0) - mn) / scale


def add_speckle_noise(data, sigma, noise):
    s, scale, mn = _minmax01(np.asarray(data, dtype=np.float64))
    return (np.clip(s + s * (noise * sigma), 0.0, 1.0) - mn) / scale


def shot_noise(data, poisson=None):
    """reference augment.py:87-94: min-max to [0,1], quantise to 1023 levels, skimage random_noise('poisson', clip=True), undo the scaling.
    skimage: vals = 2 ** ceil(log2(number of distinct values)); out = poisson(image * vals) / vals.  `poisson(lam)` defaults to numpy's
    global generator, which is what skimage draws from."""
    s, scale, mn = _minmax01(np.asarray(data, dtype=np.float64))
    q = np.floor(s * 1023) / 1023
    vals = 2 ** np.ceil(np.log2(len(np.unique(q))))
    draw = (poisson or np.random.poisson)(q * vals) / float(vals)
    return (np.clip(draw, 0.0, 1.0) - mn) / scale


def apply_gaussian_filter(data, sigma):
    """reference augment.py:113-114 = skimage.filters.gaussian(data, sigma): scipy's gaussian_filter with mode 'nearest', truncate 4; a 3-D
    array whose last axis has length 3 is taken for an RGB image by skimage (multichannel guess): no smoothing along that axis"""
    data = np.asarray(data, dtype=np.float64)
    sig = [sigma] * data.ndim
    if data.ndim == 3 and data.shape[-1] == 3:
        sig[-1] = 0
    return ndimage.gaussian_filter(data, sig, mode="nearest", truncate=4.0)


# ------------------------------------------------------------------------------------------------ imgaug (parity unpinned, see the header)
def elastic_ksize(sigma):
    """blur.py `_compute_gaussian_blur_ksize` (3.3 / 2.9 / 2.6 sigma: 99 / 97 / 95 % of the weight; at least 5) made odd as `blur_gaussian_` does"""
    k = 3.3 * sigma if sigma < 3.0 else (2.9 * sigma if sigma < 5.0 else 2.6 * sigma)
    k = int(max(k, 5))
    return k + 1 if k % 2 == 0 else k


def elastic_shift_maps(shape2d, alpha, sigma, noise):
    """geometric.py `_generate_shift_maps`: `noise` = random_state.random((2 * h_pad, w_pad)) * 2 - 1 with h_pad = h + 2 * ksize (given as
    (2, h_pad, w_pad): block 0 = dx); each block blurred by cv2.GaussianBlur((ksize, ksize), sigma) - a separable correlation with
    exp(-x^2 / 2 sigma^2) / sum on ksize taps - times alpha, padding cropped.  -> (dx, dy): displacement along axis 1, along axis 0"""
    h, w = shape2d
    k = elastic_ksize(sigma)
    noise = np.asarray(noise, dtype=np.float64)
    assert noise.shape == (2, h + 2 * k, w + 2 * k)
    xs = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
    wt = np.exp(-(xs * xs) / (2.0 * sigma * sigma))
    wt /= wt.sum()
    out = []
    for blk in noise:
        b = ndimage.correlate1d(ndimage.correlate1d(blk, wt, axis=0, mode="mirror"), wt, axis=1, mode="mirror")     # BORDER_REFLECT_101; cropped away
        out.append(b[k:k + h, k:k + w] * alpha)
    return out[0], out[1]


def elastic_apply(image, dx, dy, order):
    """geometric.py `_map_coordinates`: every channel image[:, :, c] sampled at (y - dy, x - dx), scipy map_coordinates(order, mode='nearest')
    (the reference passes mode="nearest", order 1 for the volume and 0 for truth / previous truth / mask, augment.py:151-168).
    This restates the function's SCIPY branch.  imgaug 0.4.0 sends float32 / float64 images at order 0 / 1 through cv2 instead when cv2 is
    importable (cv2.convertMaps to CV_16SC2 + cv2.remap: coordinates quantised to 1/32 pixel, fixed-point bilinear weights, cvRound for
    order 0, BORDER_REPLICATE); cv2 exists in no interpreter here, so which branch the reference's environment took is unknown - one more
    reason this augmenter is PARITY UNPINNED.  At the reference's alpha <= 5, sigma = 10 (displacements ~0.1 pixel) the branches differ by
    at most 1/64 pixel of coordinate."""
    image = np.asarray(image)
    h, w = image.shape[:2]
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    coords = np.stack([yy - dy, xx - dx])
    out = np.empty(image.shape, dtype=np.float64)
    for c in range(image.shape[2]):
        out[:, :, c] = map_coordinates(image[:, :, c].astype(np.float64), coords, order=order, mode="nearest")
    return out


def piecewise_affine_points(shape2d, jitter, nb_rows=2, nb_cols=2):
    """geometric.py PiecewiseAffine._get_transformer: a regular nb_rows x nb_cols grid of points over [0, h] x [0, w] (y, x), each moved by
    `jitter` (nb_rows * nb_cols, 2) ~ Normal(0, scale) times (h, w), destinations clipped to the image ([0, h - 1] x [0, w - 1])"""
    h, w = shape2d
    yy, xx = np.meshgrid(np.linspace(0, h, nb_rows), np.linspace(0, w, nb_cols), indexing="ij")
    src = np.stack([yy.ravel(), xx.ravel()], axis=1)
    dst = src + np.asarray(jitter, dtype=np.float64) * np.array([h, w], dtype=np.float64)
    dst[:, 0] = np.clip(dst[:, 0], 0, h - 1)
    dst[:, 1] = np.clip(dst[:, 1], 0, w - 1)
    return src, dst


def piecewise_affine_apply(image, src_yx, dst_yx, order):
    """skimage.transform.warp(image, PiecewiseAffineTransform().estimate(src (x, y), dst (x, y)), order, mode='constant', cval=0): Delaunay
    triangulation of the SOURCE points (scipy.spatial.Delaunay, as skimage does), one affine per triangle taking its source vertices to
    their destinations, output pixel (x, y) read at affine_of_its_triangle(x, y); scipy map_coordinates(mode='constant') per channel"""
    from scipy.spatial import Delaunay
    image = np.asarray(image)
    h, w = image.shape[:2]
    sxy, dxy = np.asarray(src_yx, dtype=np.float64)[:, ::-1], np.asarray(dst_yx, dtype=np.float64)[:, ::-1]
    tess = Delaunay(sxy)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    pts = np.stack([xx.ravel(), yy.ravel()], axis=1)
    simplex = tess.find_simplex(pts)
    assert (simplex >= 0).all()
    coords = np.empty_like(pts)
    for t, tri in enumerate(tess.simplices):
        S = np.hstack([sxy[tri], np.ones((3, 1))])                     # [x y 1] M = [x' y']
        M = np.linalg.solve(S, dxy[tri])
        sel = simplex == t
        coords[sel] = np.hstack([pts[sel], np.ones((int(sel.sum()), 1))]).dot(M)
    cc = np.stack([coords[:, 1].reshape(h, w), coords[:, 0].reshape(h, w)])
    out = np.empty(image.shape, dtype=np.float64)
    for c in range(image.shape[2]):
        out[:, :, c] = map_coordinates(image[:, :, c].astype(np.float64), cc, order=order, mode="constant", cval=0.0)
    return out


def coarse_dropout(data, keep_small):
    """reference augment.py:116-120: MinMaxScaler((0, 255)) over the whole array, imgaug CoarseDropout, inverse scaling.  `keep_small`: the
    Binomial(1 - rate) draw on the low-resolution grid (hs, ws, C) or (hs, ws, 1) (hs = int(h * size_percent), at least 1; size_percent drawn
    per axis); enlarged to (h, w) by nearest neighbour as cv2.resize does (source index = min(floor(i * hs / h), hs - 1)); the image is
    multiplied by it, so a dropped voxel is 0 in the scaled range = the array's minimum"""
    data = np.asarray(data, dtype=np.float64)
    h, w, c = data.shape
    hs, ws, kc = keep_small.shape
    si = np.minimum(np.floor(np.arange(h) * (hs / h)).astype(int), hs - 1)
    sj = np.minimum(np.floor(np.arange(w) * (ws / w)).astype(int), ws - 1)
    up = keep_small[si][:, sj]                                       # (h, w, kc)
    dmin, dmax = float(data.min()), float(data.max())
    rng = dmax - dmin
    scale = 255.0 / (rng if rng != 0 else 1.0)
    mn = 0.0 - dmin * scale
    scaled = (data * scale + mn) * up
    return (scaled - mn) / scale


COARSE_MIN_SIZE = 3


def coarse_dropout_grid(shape2d, size_percent, rng):
    """parameters.FromLowerResolution.draw_samples: one size_percent per axis (a list = a choice among its values, a tuple = uniform, a number
    = itself), grid = int(extent * percent), at least COARSE_MIN_SIZE = 3 per side (CoarseDropout's `min_size` default in imgaug 0.4.0)"""
    out = []
    for extent in shape2d:
        if isinstance(size_percent, list):
            sp = size_percent[int(rng.randint(len(size_percent)))]
        elif isinstance(size_percent, tuple):
            sp = rng.uniform(size_percent[0], size_percent[1])
        else:
            sp = size_percent
        out.append(max(int(extent * sp), COARSE_MIN_SIZE))
    return tuple(out)


# ------------------------------------------------------------------------------------------------ augment_data
def draw_augment_params(augment, n_dim, data_min, data_max):
    """the random draws of reference augment.py:229-291 in their order (numpy global RNG + python `random` for the flips)"""
    g = augment.get
    p = {}
    scale = list(np.random.normal(1, g("scale"), n_dim)) if g("scale") else [1, 1, 1]
    if g("iso_scale"):
        iso = np.random.uniform(1, g("iso_scale")["max"])
        if np.random.choice([True, False]):
            iso = 1 / iso
        scale[0] *= iso
        scale[1] *= iso
    p["scale"] = scale
    if g("rotate"):
        std = np.array(g("rotate"))
        p["rotate"] = np.deg2rad(np.random.uniform(low=-std, high=std, size=n_dim))
    else:
        p["rotate"] = None
    flip = g("flip")
    p["flip"] = np.arange(n_dim)[[rate > random.random() for rate in flip]] if (flip is not None and flip) else None
    if g("translate") is not None:
        t = np.random.uniform(-np.array(g("translate")), np.array(g("translate")), n_dim)
        t[-1] = np.floor(t[-1])
        p["translate"] = t
    else:
        p["translate"] = None
    if g("contrast") is not None:
        val_range = data_max - data_min
        p["contrast"] = (data_min + g("contrast")["min_factor"] * np.random.uniform(-1, 1) * val_range,
                         data_max + g("contrast")["max_factor"] * np.random.uniform(-1, 1) * val_range)
    else:
        p["contrast"] = None
    p["poisson"] = (g("poisson_noise") > np.random.random()) if g("poisson_noise") is not None else False
    p["gaussian_noise"] = (g("gaussian_noise")["prob"] > np.random.random()) if g("gaussian_noise") is not None else False
    p["speckle_noise"] = (g("speckle_noise")["prob"] > np.random.random()) if g("speckle_noise") is not None else False
    gf = g("gaussian_filter")
    if gf is not None and gf["prob"] > 0:
        p["gaussian_sigma"] = gf["max_sigma"] * np.random.random()
        p["gaussian_filter"] = gf["prob"] > np.random.random()
    else:
        p["gaussian_filter"], p["gaussian_sigma"] = False, None
    p["piecewise_affine"] = np.random.random() * g("piecewise_affine")["scale"] if g("piecewise_affine") is not None else 0
    et = g("elastic_transform")
    p["elastic"] = np.random.random() * et["alpha"] if (et is not None and et["alpha"] > 0) else 0
    im = g("intensity_multiplication")
    if im is not None:
        a, b = im
        p["intensity"] = np.random.random() * (b - a) + a
    else:
        p["intensity"] = 1
    return p


def augment_sample(data, truth, data_min, params, data_range, truth_range, prev_truth_range=None, noise=None):
    A = distort_affine(data.shape, params["flip"], params["scale"], params["rotate"], params["translate"])
    x = interpolate_affine_range(data, A, data_range, order=1, cval=data_min)
    At = distort_affine(truth.shape, params["flip"], params["scale"], params["rotate"], params["translate"])
    t = interpolate_affine_range(truth, At, truth_range, order=0, cval=0)
    pt = interpolate_affine_range(truth, At, prev_truth_range, order=0, cval=0) if prev_truth_range is not None else None
    if params["contrast"] is not None:
        x = contrast_augment(x, *params["contrast"])
    if params["intensity"] != 1:
        x = x * params["intensity"]
    if params["speckle_noise"]:
        x = add_speckle_noise(x, noise["speckle_sigma"], noise["speckle"])
    if params["gaussian_noise"]:
        x = add_gaussian_noise(x, noise["gaussian_sigma"], noise["gaussian"])
    return x, t, pt


# ------------------------------------------------------------------------------------------------ generator
class DataFileDummy:
    """reference generator.py:13-30 + pad_samples :33-57 (truth_downsample = 1)"""

    def __init__(self, data, truth, pad, patch_shape):
        self.data = [np.pad(d, pad, "constant", constant_values=d.min()) for d in data]
        self.truth = [np.pad(t, pad, "constant", constant_values=0) for t in truth]
        self.min = [float(np.min(d)) for d in self.data]
        self.max = [float(np.max(d)) for d in self.data]
        out_shape = [patch_shape[0], patch_shape[1], 1]
        padding = np.ceil(np.subtract(patch_shape, out_shape) / 2).astype(int)
        self.data = [np.pad(d, [(p, p) for p in padding], "constant", constant_values=m) for d, m in zip(self.data, self.min)]
        self.truth = [np.pad(t, [(p, p) for p in padding], "constant", constant_values=0) for t in self.truth]

        def fit(a, cv):
            p = np.ceil(np.maximum(np.subtract(patch_shape, a.shape) + 1, 0) / 2).astype(int)
            return np.pad(a, [(q, q) for q in p], "constant", constant_values=cv)

        self.data = [fit(d, m) for d, m in zip(self.data, self.min)]
        self.truth = [fit(t, 0) for t in self.truth]


def crop(a, shape, corner):
    corner = np.asarray(corner)
    assert np.all(corner >= 0) and np.all(corner + np.asarray(shape) <= np.asarray(a.shape)), "in-bounds crops only in the generator path"
    return a[corner[0]:corner[0] + shape[0], corner[1]:corner[1] + shape[1], corner[2]:corner[2] + shape[2]]


def add_data(df, index, patch_shape, augment, truth_index, truth_size, prev_truth_index=None, prev_truth_size=None, skip_blank=True):
    data, truth = df.data[index], df.truth[index]
    corner = [np.random.randint(low=0, high=h) for h in np.array(truth.shape) - np.array(patch_shape)]
    if augment is not None:
        data_range = [(s, s + n) for s, n in zip(corner, patch_shape)]
        truth_range = data_range[:2] + [(corner[2] + truth_index, corner[2] + truth_index + truth_size)]
        prev_range = data_range[:2] + [(corner[2] + prev_truth_index, corner[2] + prev_truth_index + prev_truth_size)] \
            if prev_truth_index is not None else None
        params = draw_augment_params(augment, truth.ndim, df.min[index], df.max[index])
        x, t, pt = augment_sample(data, truth, df.min[index], params, data_range, truth_range, prev_range)
    else:
        x = crop(data, patch_shape, corner)
        t = crop(truth, tuple(patch_shape[:-1]) + (truth_size,), np.array(corner) + (0, 0, truth_index))
        pt = crop(truth, tuple(patch_shape[:-1]) + (prev_truth_size,), np.array(corner) + (0, 0, prev_truth_index)) \
            if prev_truth_index is not None else None
    if pt is not None:
        x = np.concatenate([x, pt], axis=-1)
    if not skip_blank or np.any(t != 0):
        return x, t
    return None


def data_generator(df, index_list, batch_size, patch_shape, augment=None, skip_blank=True, truth_index=-1, truth_size=1,
                   prev_truth_index=None, prev_truth_size=None, is3d=False):
    """reference generator.py:222-245 with shuffle_index_list=False, categorical=False"""
    it = itertools.cycle(index_list)
    while True:
        xs, ys = [], []
        while len(xs) < batch_size:
            got = add_data(df, next(it), patch_shape, augment, truth_index, truth_size, prev_truth_index, prev_truth_size, skip_blank)
            if got is not None:
                xs.append(got[0])
                ys.append(got[1])
        x, y = np.asarray(xs), np.asarray(ys)
        if is3d:
            x, y = np.expand_dims(x, 1), np.expand_dims(y, 1)
        yield x, y


# ------------------------------------------------------------------------------------------------ permutations / TTA
def generate_permutation_keys():
    return set(itertools.product(itertools.combinations_with_replacement(range(2), 2), range(2), range(2), range(2), range(2)))


def permute_data(data, key):
    data = np.copy(data)
    (rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose = key
    if rotate_y != 0:
        data = np.rot90(data, rotate_y, axes=(1, 2))
    if flip_x:
        data = data[:, ::-1]
    if flip_y:
        data = data[:, :, ::-1]
    if flip_z:
        data = data[:, :, :, ::-1]
    return data


def reverse_permute_data(data, key):
    (rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose = key
    data = np.copy(data)
    if flip_z:
        data = data[:, :, :, ::-1]
    if flip_y:
        data = data[:, :, ::-1]
    if flip_x:
        data = data[:, ::-1]
    if rotate_y != 0:
        data = np.rot90(data, -rotate_y, axes=(1, 2))
    return data


def flip_it(a, axes):
    for ax in axes:
        a = np.flip(a, ax)
    return a


def predict_with_permutations(predict_fn, data):
    preds = [reverse_permute_data(predict_fn(permute_data(data, k)[np.newaxis])[0], k) for k in generate_permutation_keys()]
    return np.mean(preds, axis=0)


def predict_flips(patch_wise_fn, data):
    """patch_wise_fn(volume (1,X,Y,Z)) -> (X,Y,Z[,C]); the 8 subsets of axes {0,1,2} in powerset order"""
    out = []
    for r in range(4):
        for axes in itertools.combinations([0, 1, 2], r):
            d = flip_it(data, axes)
            out.append(flip_it(patch_wise_fn(np.expand_dims(d.squeeze(), 0)).squeeze(), axes).squeeze())
    return out


def predict_augment(patch_wise_fn, data, num_augments=32):
    dmax, dmin = data.max(), data.min()
    data = data.squeeze()
    preds = []
    for _ in range(num_augments):
        rng = dmax - dmin
        lo = dmin + 0.10 * np.random.uniform(-1, 1) * rng
        hi = dmax + 0.10 * np.random.uniform(-1, 1) * rng
        cur = contrast_augment(data, lo, hi)
        angle = np.random.uniform(-30, 30)
        to_flip = np.arange(0, 3)[np.random.choice([True, False], size=3)]
        to_transpose = np.random.choice([True, False])
        cur = flip_it(cur, to_flip)
        if to_transpose:
            cur = cur.transpose([1, 0, 2])
        cur = ndimage.rotate(cur, angle, order=2, reshape=False)
        p = patch_wise_fn(cur[np.newaxis, ...]).squeeze()
        p = ndimage.rotate(p, -angle)
        if to_transpose:
            p = p.transpose([1, 0, 2])
        preds.append(flip_it(p, to_flip).squeeze())
    return np.stack(preds, axis=0)
