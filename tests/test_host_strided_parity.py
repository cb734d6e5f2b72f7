"""fmri_hip.strided_parity on the CPU: the slot tables and the three data movements against a written-out stride-2 convolution
(reference isensee2017.py:51 create_convolution_block(..., strides=(2, 2, 2)); TF 'same' on even dims pads behind the volume).  The GPU
side (the parity kernels fed with these images) is tests/test_gpu_ops.py::test_stride2_conv_on_the_parity_kernels_is_exact_on_dyadic_data."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))


def _gather_conv(x, img):
    """the up-backward launch's definition on the host: low[g] = sum_{p,u} img[p][u] @ fine[2 (g + u - p) + p]   (per axis)"""
    N, D, H, W, Cin = x.shape
    Cout = img.shape[2]
    out = torch.zeros((N, D // 2, H // 2, W // 2, Cout), dtype=x.dtype)
    xp = F.pad(x.permute(0, 4, 1, 2, 3), (2, 2, 2, 2, 2, 2)).permute(0, 2, 3, 4, 1)          # fine index f -> f + 2
    for p in range(8):
        pd, ph, pw = p >> 2, (p >> 1) & 1, p & 1
        for u in range(8):
            ud, uh, uw = u >> 2, (u >> 1) & 1, u & 1
            w = img[p, u]                                                                        # [Cout][Cin]
            if not bool(w.abs().sum()):
                continue
            sd, sh, sw = 2 * (ud - pd) + pd + 2, 2 * (uh - ph) + ph + 2, 2 * (uw - pw) + pw + 2
            sl = xp[:, sd:sd + D:2, sh:sh + H:2, sw:sw + W:2]
            out += sl @ w.t()
    return out


def test_slot_tables_and_forward_image_reproduce_the_strided_convolution():
    from fmri_hip.strided_parity import StridedParity, _GATHER, _SCATTER
    sp = StridedParity("cpu")
    # every tap has its own slot, in both images; 27 of 64 used
    assert len(set(zip(sp.gp.tolist(), sp.gu.tolist()))) == 27 and len(set(zip(sp.sp.tolist(), sp.st.tolist()))) == 27
    # per axis: tap t reads input 2o + t = 2 (o + u - p) + p, and input 2g + p receives from output g + t' - 1 + p = (2g + p - t) / 2
    for t, (p, u) in _GATHER.items():
        assert 2 * (u - p) + p == t
    for t, (p, tt) in _SCATTER.items():
        assert 2 * (tt - 1 + p) + t == p
    rs = np.random.RandomState(0)
    N, D, H, W, Cin, Cout = 2, 4, 6, 8, 3, 5
    x = torch.from_numpy(rs.randn(N, D, H, W, Cin))
    w = torch.from_numpy(rs.randn(27, Cout, Cin))
    fwd = torch.zeros((8, 8, Cout, Cin), dtype=torch.float64)
    dg = torch.zeros((8, 8, Cin, Cout), dtype=torch.float64)
    sp.pack(w, fwd, dg)
    assert int((fwd.abs().sum((2, 3)) > 0).sum()) == 27
    wk = w.reshape(3, 3, 3, Cout, Cin).permute(3, 4, 0, 1, 2)
    ref = F.conv3d(F.pad(x.permute(0, 4, 1, 2, 3), (0, 1, 0, 1, 0, 1)), wk, None, stride=2).permute(0, 2, 3, 4, 1)
    assert float((_gather_conv(x, fwd) - ref).abs().max()) < 1e-12
    # the input-gradient image holds the transposed taps in the scatter twin's slots
    for t in range(27):
        assert torch.equal(dg[sp.sp[t], sp.st[t]], w[t].t())
    # unpack_wgrad reads the 27 slots back in tap order, transposed
    dwc = torch.from_numpy(rs.randn(64 * Cin * Cout))
    dw = sp.unpack_wgrad(dwc, Cout, Cin)
    v = dwc.view(8, 8, Cin, Cout)
    for t in range(27):
        assert torch.equal(dw[t], v[sp.sp[t], sp.st[t]].t())
