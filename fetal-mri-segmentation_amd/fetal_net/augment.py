"""Host side of the augmentation path: the small amount of arithmetic that decides WHAT to sample (random draws, 4x4 affines,
permutation keys); the sampling itself runs on the MI355X (fmri_hip.ops.affine_sample & co, csrc/augment.hip).

Mirrors the names of reference fetal_net/augment.py so that callers (generator, prediction) read the same:
  scale_image / translate_image / rotate_image / flip_image / distort_image   reference augment.py:14-84, :189-212
  random draws of augment_data                                                reference augment.py:229-291 (same order, same RNGs)
  generate_permutation_keys / permute_data / reverse_permute_data             reference augment.py:380-471
  contrast_augment                                                            reference augment.py:125-128 (skimage rescale_intensity restated)
"""
import itertools
import random

import numpy as np


# ---------------------------------------------------------------------------------------------------------- 4x4 affine algebra
def scale_image(affine, scale_factor):
    return np.diag(list(scale_factor) + [1]).dot(affine)


def translate_image(affine, translate_factor):
    out = np.copy(affine)
    out[0:3, 3] = out[0:3, 3] + np.asarray(translate_factor)
    return out


def rotate_image_axis(affine, angle, axis):
    s, c = np.sin(angle), np.cos(angle)
    r = np.eye(4)
    a, b = [(1, 2), (0, 2), (0, 1)][axis]
    r[a, a], r[b, b] = c, c
    if axis == 1:
        r[a, b], r[b, a] = s, -s
    else:
        r[a, b], r[b, a] = -s, s
    return r.dot(affine)


def rotate_image(affine, rotate_angles):
    out = np.copy(affine)
    for i, angle in enumerate(rotate_angles):
        if angle != 0:
            out = rotate_image_axis(out, angle, axis=i)
    return out


def flip_image(affine, axis):
    out = np.copy(affine)
    for ax in axis:
        out = rotate_image_axis(out, np.deg2rad(180), axis=int(ax))     # the reference "flips" by a half turn about the axis
    return out


def distort_image(data, affine, flip_axis=None, scale_factor=None, rotate_factor=None, translate_factor=None):
    """-> (data, affine'): centre the volume, flip / scale / rotate, move back, translate"""
    centre = np.array(data.shape) / 2
    affine = translate_image(affine, -centre)
    if flip_axis is not None:
        affine = flip_image(affine, flip_axis)
    if scale_factor is not None:
        affine = scale_image(affine, scale_factor)
    if rotate_factor is not None:
        affine = rotate_image(affine, rotate_factor)
    affine = translate_image(affine, +centre)
    if translate_factor is not None:
        affine = translate_image(affine, translate_factor)
    return data, affine


# ---------------------------------------------------------------------------------------------------------- random draws
def random_scale_factor(n_dim=3, mean=1, std=0.25):
    return np.random.normal(mean, std, n_dim)


def random_translate_factor(n_dim=3, min=0, max=7):
    return np.random.uniform(min, max, n_dim)


def random_rotation_angle(n_dim=3, mean=0, std=5):
    return np.random.uniform(low=mean - np.array(std), high=mean + np.array(std), size=n_dim)


def random_boolean():
    return np.random.choice([True, False])


def random_flip_dimensions(n_dim, flip_factor):
    return np.arange(n_dim)[[flip_rate > random.random() for flip_rate in flip_factor]]


def draw_augment_parameters(augment, n_dim, data_min, data_max):
    """Every random number augment_data draws before it touches the data, in its order (numpy's global generator; python's `random`
    for the flips), so that a seeded run picks the same transformation as the reference.  Draws belonging to augmenters this
    package does not apply (poisson, gaussian filter, piecewise affine, elastic, coarse dropout) are still consumed."""
    g = augment.get
    p = {}
    scale_factor = list(random_scale_factor(n_dim, std=g("scale"))) if g("scale") else [1, 1, 1]
    if g("iso_scale"):
        iso = np.random.uniform(1, g("iso_scale")["max"])
        if random_boolean():
            iso = 1 / iso
        scale_factor[0] *= iso
        scale_factor[1] *= iso
    p["scale_factor"] = scale_factor
    p["rotate_factor"] = np.deg2rad(random_rotation_angle(n_dim, std=g("rotate"))) if g("rotate") else None
    flip = g("flip")
    p["flip_axis"] = random_flip_dimensions(n_dim, flip) if (flip is not None and flip) else None
    if g("translate") is not None:
        t = random_translate_factor(n_dim, -np.array(g("translate")), np.array(g("translate")))
        t[-1] = np.floor(t[-1])                                    # whole slices only
        p["translate_factor"] = t
    else:
        p["translate_factor"] = None
    if g("contrast") is not None:
        val_range = data_max - data_min
        lo = data_min + g("contrast")["min_factor"] * np.random.uniform(-1, 1) * val_range
        hi = data_max + g("contrast")["max_factor"] * np.random.uniform(-1, 1) * val_range
        p["contrast"] = (lo, hi)
    else:
        p["contrast"] = None
    p["apply_poisson_noise"] = (g("poisson_noise") > np.random.random()) if g("poisson_noise") is not None else False
    p["apply_gaussian_noise"] = (g("gaussian_noise")["prob"] > np.random.random()) if g("gaussian_noise") is not None else False
    p["apply_speckle_noise"] = (g("speckle_noise")["prob"] > np.random.random()) if g("speckle_noise") is not None else False
    gf = g("gaussian_filter")
    if gf is not None and gf["prob"] > 0:
        p["gaussian_sigma"] = gf["max_sigma"] * np.random.random()
        p["apply_gaussian_filter"] = gf["prob"] > np.random.random()
    else:
        p["apply_gaussian_filter"], p["gaussian_sigma"] = False, None
    p["piecewise_affine_scale"] = np.random.random() * g("piecewise_affine")["scale"] if g("piecewise_affine") is not None else 0
    et = g("elastic_transform")
    p["elastic_transform_scale"] = np.random.random() * et["alpha"] if (et is not None and et["alpha"] > 0) else 0
    im = g("intensity_multiplication")
    if im is not None:
        a, b = im
        p["intensity_multiplication"] = np.random.random() * (b - a) + a
    else:
        p["intensity_multiplication"] = 1
    p["coarse_dropout"] = g("coarse_dropout") is not None
    return p


# ---------------------------------------------------------------------------------------------------------- intensity (host form, TTA)
def contrast_augment(data, min_per, max_per):
    data = np.asarray(data, dtype=np.float64)
    omin, omax = float(data.min()), float(data.max())
    out = np.clip(data, min_per, max_per)
    if min_per != max_per:
        return (out - min_per) / (max_per - min_per) * (omax - omin) + omin
    return np.clip(out, omin, omax)


# ---------------------------------------------------------------------------------------------------------- the 48 cube isometries
def generate_permutation_keys():
    """keys ((rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose); as in the reference only rotate_y and the flips act"""
    return set(itertools.product(itertools.combinations_with_replacement(range(2), 2), range(2), range(2), range(2), range(2)))


def random_permutation_key():
    return random.choice(list(generate_permutation_keys()))


def permute_data(data, key):
    """data (n_modalities, x, y, z)"""
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


def random_permutation_x_y(x_data, y_data):
    key = random_permutation_key()
    return permute_data(x_data, key), permute_data(y_data, key)


def reverse_permutation_key(key):
    return tuple(-r for r in key[0]), key[1], key[2], key[3], key[4]


def reverse_permute_data(data, key):
    (rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose = reverse_permutation_key(key)
    data = np.copy(data)
    if flip_z:
        data = data[:, :, :, ::-1]
    if flip_y:
        data = data[:, :, ::-1]
    if flip_x:
        data = data[:, ::-1]
    if rotate_y != 0:
        data = np.rot90(data, rotate_y, axes=(1, 2))
    return data
