"""Slow, independent numpy restatement of the tensor ops (test infrastructure; cross-checks oracle/unet_oracle.py).

Plain shifted-slice accumulation in float64 — no torch, no FFT, no im2col — so that the torch-CPU oracle and this file
share no code path.  Small shapes only.  Semantics follow Keras/TF as used by reference
fetal_net/model/unet3d/unet.py:89-138 (Conv3D 'same', MaxPooling3D(2), UpSampling3D(2), Conv3DTranspose k2 s2).
"""
import numpy as np


def conv_same(x, k, b=None):
    """x (N,Cin,*sp) ; k Keras layout (k..., Cin, Cout), odd k, stride 1, zero 'same' padding."""
    nd = x.ndim - 2
    ks = k.shape[:nd]
    pads = [(0, 0), (0, 0)] + [((kk - 1) // 2, kk // 2) for kk in ks]
    xp = np.pad(x.astype(np.float64), pads)
    N, sp = x.shape[0], x.shape[2:]
    out = np.zeros((N, k.shape[-1]) + tuple(sp), np.float64)
    for tap in np.ndindex(*ks):
        sl = tuple(slice(t, t + s) for t, s in zip(tap, sp))
        xs = xp[(slice(None), slice(None)) + sl]                 # (N,Cin,*sp)
        out += np.einsum("nc...,co->no...", xs, k[tap].astype(np.float64))
    if b is not None:
        out += b.reshape((1, -1) + (1,) * nd)
    return out


def maxpool2(x):
    nd = x.ndim - 2
    sp = x.shape[2:]
    shp = x.shape[:2] + tuple(v for s in sp for v in (s // 2, 2))
    xr = x.reshape(shp)
    for i in range(nd):
        xr = xr.max(axis=3 + i)          # after each reduction the next '2' axis sits at 3+i
    return xr


def upsample2(x):
    for ax in range(2, x.ndim):
        x = np.repeat(x, 2, axis=ax)
    return x


def deconv_k2s2(x, k, b=None):
    """Keras Conv3DTranspose kernel (2,2,2,Cout,Cin), stride 2, 'valid': out[2i+a] = sum_ci x[i,ci] * k[a,:,ci]."""
    nd = x.ndim - 2
    N, sp = x.shape[0], x.shape[2:]
    cout = k.shape[-2]
    out = np.zeros((N, cout) + tuple(2 * s for s in sp), np.float64)
    for tap in np.ndindex(*k.shape[:nd]):
        sl = tuple(slice(t, None, 2) for t in tap)
        out[(slice(None), slice(None)) + sl] = np.einsum("nc...,oc->no...", x.astype(np.float64), k[tap].astype(np.float64))
    if b is not None:
        out += b.reshape((1, -1) + (1,) * nd)
    return out


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))
