"""Weight algebra of the folded transposed conv (fmri_hip/deconv_fold.py) on the CPU in fp64: Deconvolution3D(k 2, s 2) -> concatenate ->
Conv3D(3x3x3, 'same') written out with torch (= the reference's three Keras layers, unet.py:132-138, :61, :102) against the parity-form
evaluation with effective filters and per-border-class bias; and the gradients chained back through the effective filters against autograd
of the written-out form."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))


def _reference(x_low, skip, w3, wt, b3, bt, cmid):
    """x_low [N, Cin, d, h, w], skip [N, Cs, 2d, 2h, 2w]; w3 [27, Cout, Cmid + Cs], wt [8, Cmid, Cin] (engine layouts)"""
    Cin = wt.shape[2]
    kt = wt.view(2, 2, 2, cmid, Cin).permute(4, 3, 0, 1, 2)              # conv_transpose3d weight: (Cin, Cmid, 2, 2, 2)
    up = F.conv_transpose3d(x_low, kt, bt, stride=2)
    k3 = w3.view(3, 3, 3, w3.shape[1], w3.shape[2]).permute(3, 4, 0, 1, 2)
    return F.conv3d(torch.cat([up, skip], 1), k3, b3, padding=1)


def _folded(x_low, skip, weff, bias27, w3, cmid):
    N, Cin, d, h, w = x_low.shape
    Cout = weff.shape[2]
    y = torch.zeros((N, Cout, 2 * d, 2 * h, 2 * w), dtype=x_low.dtype)
    xp = F.pad(x_low, (1, 1, 1, 1, 1, 1))
    for p in range(8):
        pd, ph, pw = p >> 2, (p >> 1) & 1, p & 1
        k = weff[p].view(2, 2, 2, Cout, Cin).permute(3, 4, 0, 1, 2)
        # parity 0 reads low-res voxels (g-1, g) = padded (g, g+1); parity 1 reads (g, g+1) = padded (g+1, g+2)
        win = xp[:, :, pd:pd + d + 1, ph:ph + h + 1, pw:pw + w + 1]
        y[:, :, pd::2, ph::2, pw::2] = F.conv3d(win, k)
    k3s = w3[:, :, cmid:].reshape(3, 3, 3, Cout, -1).permute(3, 4, 0, 1, 2)
    y = y + F.conv3d(skip, k3s, None, padding=1)
    D, H, W = 2 * d, 2 * h, 2 * w
    cls = lambda n: torch.tensor([0 if i == 0 else (2 if i == n - 1 else 1) for i in range(n)])
    c = (cls(D)[:, None, None] * 3 + cls(H)[None, :, None]) * 3 + cls(W)[None, None, :]
    return y + bias27[c].permute(3, 0, 1, 2)[None]


def test_folded_transposed_conv_equals_the_three_layers_and_chains_its_gradients():
    from fmri_hip.deconv_fold import DeconvFold
    torch.manual_seed(3)
    f64 = torch.float64
    N, Cin, cmid, Cs, Cout, d, h, w = 2, 5, 4, 3, 6, 3, 2, 4
    x_low = torch.randn(N, Cin, d, h, w, dtype=f64)
    skip = torch.randn(N, Cs, 2 * d, 2 * h, 2 * w, dtype=f64)
    w3 = (torch.randn(27, Cout, cmid + Cs, dtype=f64) * 0.2).requires_grad_(True)
    wt = (torch.randn(8, cmid, Cin, dtype=f64) * 0.3).requires_grad_(True)
    b3 = torch.randn(Cout, dtype=f64).requires_grad_(True)
    bt = torch.randn(cmid, dtype=f64).requires_grad_(True)
    fold = DeconvFold("cpu")
    y_ref = _reference(x_low, skip, w3, wt, b3, bt, cmid)
    weff, bias27 = fold.effective(w3.detach(), wt.detach(), b3.detach(), bt.detach(), cmid)
    y_fold = _folded(x_low, skip, weff, bias27, w3.detach(), cmid)
    assert float((y_fold - y_ref).abs().max()) < 1e-12 * float(y_ref.abs().max())
    # gradients: autograd of the written-out form vs the chain through the effective filters
    dy = torch.randn_like(y_ref)
    (y_ref * dy).sum().backward()
    weff_l = weff.clone().requires_grad_(True)
    b27_l = bias27.clone().requires_grad_(True)
    (_folded(x_low, skip, weff_l, b27_l, w3.detach(), cmid) * dy).sum().backward()
    s27 = b27_l.grad                                                   # per-border-class sums of dy = what fmri_border_class_sums produces
    D, H, W = 2 * d, 2 * h, 2 * w
    assert torch.allclose(s27.sum(0), dy.sum((0, 2, 3, 4)))
    dw3u, dwt, dbt = fold.chain(weff_l.grad, w3.detach(), wt.detach(), bt.detach(), cmid, s27)
    for got, want, name in ((dw3u, w3.grad[:, :, :cmid], "dW3 (up-sampled columns)"), (dwt, wt.grad, "dWt"), (dbt, bt.grad, "dbt")):
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 1e-9, (name, err)
    assert torch.allclose(s27.sum(0), b3.grad)


def test_border_class_tables():
    from fmri_hip.deconv_fold import _tables
    tq, aq, present = _tables()
    # parity 0 per axis: tap 0 -> neighbour g-1 (t' 0) through Wt[1]; taps 1, 2 -> g (t' 1) through Wt[0], Wt[1]
    assert tq[0, 0] == 0 and aq[0, 0] == 7 and tq[0, 13] == 7 and aq[0, 13] == 0 and tq[0, 26] == 7 and aq[0, 26] == 7
    # parity 1 per axis: taps 0, 1 -> g (t' 0) through Wt[0], Wt[1]; tap 2 -> g+1 (t' 1) through Wt[0]
    assert tq[7, 0] == 0 and aq[7, 0] == 0 and tq[7, 13] == 0 and aq[7, 13] == 7 and tq[7, 26] == 7 and aq[7, 26] == 0
    assert present[13].sum() == 27 and present[0].sum() == 8 and present[26].sum() == 8 and present[4].sum() == 18 and present[1].sum() == 12   # corner, face, edge
