"""The reference's utils/utils.py names that this package has a counterpart for (reference fetal_net/utils/utils.py:18-36, :76-79, :100-113):
pickles, image objects, and the affine resampling the host augmentation is built on.  Image resizing / re-orientation (nilearn, SimpleITK based
in the reference) is outside the hot-path scope."""
import os

import numpy as np

from ..data import pickle_dump, pickle_load  # noqa: F401
from .nifti import NiftiImage, get_image, load_nifti  # noqa: F401


def read_img(in_file):
    """reference :76-79 (nib.load): -> NiftiImage with the stored data (scaled when the header sets a slope) and the file's affine"""
    data, affine = load_nifti(os.path.abspath(in_file), return_affine=True)
    return NiftiImage(data, affine)


def get_affine(in_file):
    """reference :34-35"""
    return read_img(in_file).affine


def interpolate_affine_coords(data, affine, coords, mode='constant', order=0, cval=0):
    """data sampled at affine . (i, j, k, 1) for every (i, j, k) of the grid coords[0] x coords[1] x coords[2] (reference :100-108:
    nibabel's apply_affine + scipy map_coordinates)"""
    from scipy.ndimage import map_coordinates
    affine = np.asarray(affine, dtype=np.float64)
    grid = np.array(np.meshgrid(*coords, indexing='ij'), dtype=np.float64)                       # (3, a, b, c)
    src = np.tensordot(affine[:3, :3], grid, axes=([1], [0])) + affine[:3, 3].reshape(3, 1, 1, 1)
    return map_coordinates(data, src, mode=mode, order=order, cval=cval)


def interpolate_affine_range(data, affine, ranges, mode='constant', order=0, cval=0):
    """reference :111-113"""
    return interpolate_affine_coords(data, affine, coords=[range(s, e) for s, e in ranges], mode=mode, order=order, cval=cval)
