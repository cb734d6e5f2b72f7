"""Volume pre-filters a data file can be built with (reference fetal_net/preprocess.py:5-27; selected by config["preproc"] in the reference's
config_utils.py:176-181 and handed to data.write_data_to_file as `preproc`).  Host-side scipy, applied once per volume when the file is written."""
from scipy import ndimage


def norm_minmax(d):
    """linear map of the volume's range onto [-1, 1]"""
    lo, hi = d.min(), d.max()
    return -1 + 2 * (d - lo) / (hi - lo)


def laplace(d):
    return ndimage.laplace(d)


def laplace_norm(d):
    return norm_minmax(laplace(d))


def grad(d):
    """Gaussian gradient magnitude, sigma 1 voxel on every axis"""
    return ndimage.gaussian_gradient_magnitude(d, sigma=(1, 1, 1))


def grad_norm(d):
    return norm_minmax(grad(d))
