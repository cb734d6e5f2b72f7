"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) on numpy - the host restatement of the counter-based generator inside
csrc/augment.hip's *_rng kernels.  Test infrastructure: pinned to Random123's known-answer vectors by tests/test_philox_ref.py, used by
tests/test_gpu_augment.py to reproduce on the host the draws a kernel made."""
import numpy as np

_M0, _M1, _MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
_W0, _W1, _S32 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """counter words c0..c3 (arrays or scalars, broadcast together), key words k0, k1 -> four uint32 arrays"""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3)])
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = (p1 >> _S32) ^ c1 ^ k0, p1 & _MASK, (p0 >> _S32) ^ c3 ^ k1, p0 & _MASK
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return [v.astype(np.uint32) for v in (c0, c1, c2, c3)]


def u01(word):
    """the kernels' uniform on [0, 1): the top 24 bits"""
    return (np.asarray(word, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) / 16777216.0
