"""Host-side tile extraction for the sliding-window path when the model is a foreign object (the device path gathers its tiles with
fmri_tile_gather).  Same contract as the reference helper of the same name (fetal_net/utils/patches.py:57-91): a corner-indexed patch of
the last three axes, parts that fall outside the volume replicate the nearest edge voxel."""
import numpy as np


def get_patch_from_3d_data(data, patch_shape, patch_index):
    """Clamped-index gather: per axis the wanted coordinates corner .. corner + size - 1 are clipped into [0, dim - 1] - exactly what an
    edge-mode pad followed by a slice produces, without building the padded volume.  Leading axes (channels) are kept.  Always a copy."""
    data = np.asarray(data)
    dims = data.shape[-3:]
    coords = [np.clip(np.arange(int(c), int(c) + int(n)), 0, d - 1) for c, n, d in zip(patch_index, patch_shape, dims)]
    if len(coords) != 3:
        raise ValueError("patch_index / patch_shape must address the last three axes")
    return data[(Ellipsis,) + np.ix_(*coords)]


def get_random_patch_index(image_shape, patch_shape):
    """uniform random corner such that the patch lies inside the image; one np.random.choice draw per axis (the draw order is part of the
    seeded-generator contract, tests/golden/augment_golden.*)"""
    return tuple(int(np.random.choice(int(i) - int(p) + 1)) for i, p in zip(image_shape, patch_shape))
