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


# ---------------------------------------------------------------------------------------------- the reference's patch-grid helpers
# reference fetal_net/utils/patches.py:8-55, :75-152.  The prediction path of this package (and of the reference: prediction.py:188-210 builds its
# own index list and accumulates in place) does not call them; they are here under their names for scripts written against the reference module.
def get_set_of_patch_indices(start, stop, step):
    """all corners start, start + step, ... < stop per axis, x slowest (reference :37-39)"""
    axes = [np.arange(s, e, st) for s, e, st in zip(start, stop, step)]
    return np.asarray(np.stack(np.meshgrid(*axes, indexing="ij"), axis=0).reshape(3, -1).T, dtype=int)


def compute_patch_indices(image_shape, patch_size, overlap, start=None):
    """corner grid that covers the image with patches overlapping by `overlap` voxels; without `start` the grid is centred: the overhang is split
    over both ends, so the first corners are negative (reference :8-19)"""
    image_shape, patch_size = np.asarray(image_shape), np.asarray(patch_size)
    if isinstance(overlap, int):
        overlap = np.asarray([overlap] * len(image_shape))
    if start is None:
        n_patches = np.ceil(image_shape / (patch_size - overlap))
        overflow = (patch_size - overlap) * n_patches - image_shape + overlap
        start = -np.ceil(overflow / 2)
    elif isinstance(start, int):
        start = np.asarray([start] * len(image_shape))
    return get_set_of_patch_indices(start, image_shape + start, patch_size - overlap)


def get_random_nd_index(index_max):
    """reference :53-54 (one np.random.choice per axis)"""
    return tuple(int(np.random.choice(int(m) + 1)) for m in index_max)


def fix_out_of_bound_patch_attempt(data, patch_shape, patch_index, ndim=3):
    """edge-pad `data` so that the patch at `patch_index` lies inside it -> (padded data, shifted index) (reference :75-91); get_patch_from_3d_data
    above reaches the same patch with clamped indices, without the padded copy"""
    data, patch_index, patch_shape = np.asarray(data), np.asarray(patch_index), np.asarray(patch_shape)
    image_shape = np.asarray(data.shape[-ndim:])
    pad_before = np.abs((patch_index < 0) * patch_index)
    pad_after = np.abs(((patch_index + patch_shape) > image_shape) * ((patch_index + patch_shape) - image_shape))
    pad_args = [[0, 0]] * (data.ndim - ndim) + np.stack([pad_before, pad_after], axis=1).tolist()
    return np.pad(data, pad_args, mode="edge"), patch_index + pad_before


def reconstruct_from_patches(patches, patch_indices, data_shape, default_value=0):
    """the array of shape `data_shape` = (X, Y, Z, ...) rebuilt from patches (px, py, pz, ...) at the given corners; parts of a patch outside the
    array are dropped, overlapping patches averaged; every voxel must be covered (reference :94-152, which asserts the same and never uses
    default_value)"""
    data = np.zeros(data_shape, dtype=np.float64)
    count = np.zeros(data_shape, dtype=np.int64)
    image_shape = np.asarray(data_shape[:3])
    for patch, index in zip(patches, patch_indices):
        patch, index = np.asarray(patch), np.asarray(index, dtype=int)
        lo = np.maximum(-index, 0)                                   # first patch voxel inside the array
        hi = np.minimum(np.asarray(patch.shape[:3]), image_shape - index)
        if np.any(hi <= lo):
            continue
        dst = tuple(slice(int(i + a), int(i + b)) for i, a, b in zip(index, lo, hi))
        data[dst] += patch[tuple(slice(int(a), int(b)) for a, b in zip(lo, hi))]
        count[dst] += 1
    assert np.all(count > 0)
    return data / count
