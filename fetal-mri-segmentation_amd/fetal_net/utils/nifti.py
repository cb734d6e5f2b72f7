"""Minimal NIfTI-1 writer/reader (single-file .nii / .nii.gz, identity affine) so that run_validation_case can emit the
reference's output files without nibabel (reference prediction.py:302-329 via utils/utils.py:18-21 `get_image`)."""
import gzip
import struct

import numpy as np

_DT = {np.dtype(np.uint8): (2, 8), np.dtype(np.int16): (4, 16), np.dtype(np.int32): (8, 32), np.dtype(np.float32): (16, 32),
       np.dtype(np.float64): (64, 64), np.dtype(np.int8): (256, 8), np.dtype(np.uint16): (512, 16)}


def save_nifti(data, path, affine=None):
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype not in _DT:
        data = data.astype(np.float32)
    code, bits = _DT[data.dtype]
    affine = np.eye(4) if affine is None else np.asarray(affine, np.float64)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, code)
    struct.pack_into("<h", hdr, 72, bits)
    struct.pack_into("<8f", hdr, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)                  # vox_offset
    struct.pack_into("<f", hdr, 112, 1.0)                    # scl_slope
    struct.pack_into("<h", hdr, 252, 0)                      # qform_code
    struct.pack_into("<h", hdr, 254, 2)                      # sform_code = aligned
    struct.pack_into("<4f", hdr, 280, *affine[0])
    struct.pack_into("<4f", hdr, 296, *affine[1])
    struct.pack_into("<4f", hdr, 312, *affine[2])
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + b"\x00\x00\x00\x00" + np.asfortranarray(data).tobytes(order="F")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(payload)
    return path


def load_nifti(path, return_affine=False, scaled=True):
    """Read a single-file NIfTI-1 image (.nii / .nii.gz), either byte order.  With `scaled` the stored values are mapped through
    scl_slope / scl_inter when a slope is set (what nibabel's get_fdata returns); otherwise the stored dtype is kept.
    `return_affine`: also return the 4x4 voxel-to-world matrix (sform when sform_code > 0, else the pixdim scaling)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    en = "<" if struct.unpack_from("<i", raw, 0)[0] == 348 else ">"
    if struct.unpack_from(en + "i", raw, 0)[0] != 348:
        raise ValueError("%s is not a NIfTI-1 file (sizeof_hdr != 348)" % path)
    if raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError("%s: unknown NIfTI magic %r" % (path, raw[344:348]))
    if raw[344:347] == b"ni1":
        raise ValueError("%s: two-file NIfTI (.hdr/.img) is not supported" % path)
    dim = struct.unpack_from(en + "8h", raw, 40)
    code = struct.unpack_from(en + "h", raw, 70)[0]
    pixdim = struct.unpack_from(en + "8f", raw, 76)
    off = int(struct.unpack_from(en + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(en + "2f", raw, 112)
    by_code = {v[0]: k for k, v in _DT.items()}
    if code not in by_code:
        raise ValueError("%s: unsupported NIfTI datatype code %d" % (path, code))
    dt = by_code[code].newbyteorder(en)
    shape = dim[1:1 + dim[0]]
    data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    if scaled and slope not in (0.0, 1.0) or (scaled and slope != 0.0 and inter != 0.0):
        data = data.astype(np.float64) * slope + inter
    elif data.dtype.byteorder not in ("=", "|") and en == ">":
        data = data.astype(data.dtype.newbyteorder("="))
    if not return_affine:
        return data
    affine = np.eye(4)
    if struct.unpack_from(en + "h", raw, 254)[0] > 0:
        for r, o in enumerate((280, 296, 312)):
            affine[r] = struct.unpack_from(en + "4f", raw, o)
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = pixdim[1:4]
    return data, affine
