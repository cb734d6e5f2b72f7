#!/usr/bin/env python3
"""Decoder 'a' layers of BASELINE config 2 (up-sample + concat + conv): the 27-tap kernels against the parity form
(fmri_conv3d_upcat_fwd / _dgrad).  Forward and input-gradient, ms per launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops

LAYERS = [("dec2a", 512, 256, 256, 16, 32, 32), ("dec1a", 256, 128, 128, 32, 64, 64), ("dec0a", 128, 64, 64, 64, 128, 128)]
N = 4


def timeit(f, iters=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = [0, 0, 0, 0, 0, 0]
for name, C0, C1, Cout, D, H, W in LAYERS:
    bf = torch.bfloat16
    xl = torch.randn((N, D // 2, H // 2, W // 2, C0), device="cuda").to(bf)
    xs = torch.randn((N, D, H, W, C1), device="cuda").to(bf)
    w = (torch.randn((27, Cout, C0 + C1), device="cuda") * 0.05)
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=bf)
    wf, wd = torch.empty((27, Cout, C0 + C1), device="cuda", dtype=bf), torch.empty((27, C0 + C1, Cout), device="cuda", dtype=bf)
    ops.pack_weights(w, wf, wd)
    up_f, up_d = torch.empty((8, 8, Cout, C0), device="cuda", dtype=bf), torch.empty((8, 8, C0, Cout), device="cuda", dtype=bf)
    sk_f, sk_d = torch.empty((27, Cout, C1), device="cuda", dtype=bf), torch.empty((27, C1, Cout), device="cuda", dtype=bf)
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
    dy = torch.randn((N, D, H, W, Cout), device="cuda").to(bf)
    cat = torch.empty((N, D, H, W, C0 + C1), device="cuda", dtype=bf)
    dxl, dxs = torch.empty_like(xl), torch.empty_like(xs)
    t_f27 = timeit(lambda: ops.conv3d_fwd(xl, xs, wf, b, y, up0=True))
    t_fup = timeit(lambda: ops.conv3d_upcat_fwd(xl, xs, up_f, sk_f, b, y))

    def old_bwd():
        ops.conv3d_dgrad(dy, wd, cat)
        ops.upsample_bwd(cat, dxl, 0, xmask=xl)

    t_b27 = timeit(old_bwd)
    t_bup = timeit(lambda: ops.conv3d_upcat_dgrad(dy, up_d, sk_d, xl, None, dxl, dxs))
    dw, db = torch.zeros((27, Cout, C0 + C1), device="cuda"), torch.zeros(Cout, device="cuda")
    scratch = torch.empty(64 * Cout * C0, device="cuda")
    nws = ops.conv3d_wgrad_workspace_bytes(C0, C1, Cout, N, D, H, W, bf)
    ws = torch.empty(max(nws // 4, 1), device="cuda") if nws else None
    t_w27 = timeit(lambda: ops.conv3d_wgrad(xl, xs, dy, dw, db, up0=True, workspace=ws))
    t_wup = timeit(lambda: ops.conv3d_upcat_wgrad(xl, xs, dy, dw, db, scratch, workspace=ws))
    print("%s  fwd %.3f -> %.3f ms   dgrad(+upsample_bwd) %.3f -> %.3f ms   wgrad %.3f -> %.3f ms" % (name, t_f27, t_fup, t_b27, t_bup, t_w27, t_wup))
    for i, v in enumerate((t_f27, t_fup, t_b27, t_bup, t_w27, t_wup)):
        tot[i] += v
print("total fwd %.3f -> %.3f   dgrad %.3f -> %.3f   wgrad %.3f -> %.3f" % tuple(tot))
