"""Patch extraction helpers on the live prediction path (reference fetal_net/utils/patches.py:57-91)."""
import numpy as np


def get_patch_from_3d_data(data, patch_shape, patch_index):
    """corner-indexed patch of the last three axes; out-of-range parts replicate the edge."""
    patch_index = np.asarray(patch_index, dtype=np.int16)
    patch_shape = np.asarray(patch_shape)
    image_shape = data.shape[-3:]
    if np.any(patch_index < 0) or np.any((patch_index + patch_shape) > image_shape):
        data, patch_index = fix_out_of_bound_patch_attempt(data, patch_shape, patch_index)
    i, s = patch_index, patch_shape
    return data[..., i[0]:i[0] + s[0], i[1]:i[1] + s[1], i[2]:i[2] + s[2]]


def fix_out_of_bound_patch_attempt(data, patch_shape, patch_index, ndim=3):
    image_shape = np.asarray(data.shape[-ndim:])
    pad_before = np.abs((patch_index < 0) * patch_index)
    pad_after = np.abs(((patch_index + patch_shape) > image_shape) * ((patch_index + patch_shape) - image_shape))
    pad_args = np.stack([pad_before, pad_after], axis=1).tolist()
    pad_args = [[0, 0]] * (data.ndim - len(pad_args)) + pad_args
    return np.pad(data, pad_args, mode="edge"), patch_index + pad_before


def get_random_nd_index(index_max):
    return tuple([np.random.choice(index_max[index] + 1) for index in range(len(index_max))])


def get_random_patch_index(image_shape, patch_shape):
    return get_random_nd_index(np.subtract(image_shape, patch_shape))
