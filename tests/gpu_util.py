"""Helpers shared by the -m gpu parity tests (checker side: torch-CPU fp64 restatements from oracle/)."""
import numpy as np
import torch
import torch.nn.functional as F


def to_ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


def to_ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def keras_kernel_from_packed(w):
    """[27][Cout][Cin] -> torch conv weight (Cout, Cin, 3,3,3)"""
    Cout, Cin = w.shape[1], w.shape[2]
    return w.reshape(3, 3, 3, Cout, Cin).permute(3, 4, 0, 1, 2).contiguous()


def ref_concat_input(src0, src1, up0, planar=False):
    """NDHWC cpu fp64 tensors -> NCDHW concat (up first)"""
    a = to_ncdhw(src0)
    if up0:
        for ax in ((3, 4) if planar else (2, 3, 4)):
            a = torch.repeat_interleave(a, 2, dim=ax)
    if src1 is not None:
        a = torch.cat([a, to_ncdhw(src1)], dim=1)
    return a


def planar_kernel(k):
    """zero the kd != 1 planes of a torch conv weight (Cout,Cin,3,3,3): the 2-D slice semantics"""
    k = k.clone()
    k[:, :, 0] = 0
    k[:, :, 2] = 0
    return k


def ref_conv_fwd(src0, src1, up0, w, bias, act, planar=False):
    x = ref_concat_input(src0, src1, up0, planar)
    k = keras_kernel_from_packed(w)
    y = F.conv3d(x, planar_kernel(k) if planar else k, bias, padding=1)
    if act == 1:
        y = F.relu(y)
    return to_ndhwc(y)


def rnd(shape, seed, dtype, scale=1.0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(*shape, generator=g) * scale).to(dtype)
    return t.to(device)


def f64(t):
    return t.detach().to("cpu").to(torch.float64)


def assert_close(got, ref, rtol, atol_rel, what=""):
    got, ref = f64(got), f64(ref)
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    tol = atol_rel * scale + rtol * ref.abs()
    bad = err > tol
    if bool(bad.any()):
        idx = torch.nonzero(bad)[:5].tolist()
        raise AssertionError("%s: %d/%d elements off; max err %.3e (scale %.3e); first idx %s got %s ref %s" % (
            what, int(bad.sum()), bad.numel(), float(err.max()), scale, idx,
            [float(got[tuple(i)]) for i in idx], [float(ref[tuple(i)]) for i in idx]))
    return float(err.max()) / scale
