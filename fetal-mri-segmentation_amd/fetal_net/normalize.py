"""Data-file normalisation under the reference's module name (fetal_net/normalize.py:64-92).  The implementations live in fetal_net.data beside
write_data_to_file, which applies them; the image-cropping helpers of the reference's module (nilearn / nibabel based, used by its
stand-alone pre-cropping scripts only) are not part of this package."""
import numpy as np

from .data import normalize_data_storage, normalize_data_storage_each  # noqa: F401


def normalize_data(data, mean, std):
    """(data - mean) / std, in place like the reference (normalize.py:64-67)"""
    data -= mean
    data /= std
    return data
