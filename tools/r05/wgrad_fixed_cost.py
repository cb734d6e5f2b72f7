#!/usr/bin/env python3
"""Fixed cost (ramp + flush + tail) against per-unit cost of the kd-sharing weight-gradient launch: the same layer at N = 1, 2, 4, 8 samples
(units per workgroup scale with N at a fixed workgroup count); t(N) = fixed + N * per_sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops
for name, C0, Cout, D, H, W in (("dec0b", 64, 64, 64, 128, 128), ("enc0b", 32, 64, 64, 128, 128), ("dec1b", 128, 128, 32, 64, 64)):
    ts, ghz = {}, {}
    for N in (1, 2, 4, 8):
        x = torch.randn((N, D, H, W, C0), device="cuda").to(torch.bfloat16)
        dy = torch.randn((N, D, H, W, Cout), device="cuda").to(torch.bfloat16)
        dw = torch.zeros((27, Cout, C0), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        f = lambda: ops.conv3d_wgrad(x, None, dy, dw, db)
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(int(60 / N)):          # ~1 s of back-to-back launches first: the clock of THIS load
            f()
        s0, s1 = torch.zeros(16, dtype=torch.int64, device="cuda"), torch.zeros(16, dtype=torch.int64, device="cuda")
        ops.clock_stamp(s0)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        ops.clock_stamp(s1)
        torch.cuda.synchronize()
        ts[N] = e0.elapsed_time(e1) / 20
        ghz[N] = ops.clock_ghz(s0, s1)[0]
    per = (ts[8] - ts[2]) / 6
    fixed = ts[4] - 4 * per
    print("%-6s ms at N=1,2,4,8: %s | per sample %.4f ms, fixed %.4f ms (%.1f %% of the N = 4 launch)" % (
        name, " ".join("%.4f" % ts[n] for n in (1, 2, 4, 8)), per, fixed, 100 * fixed / ts[4]), "| clock GHz", " ".join("%.2f" % ghz[n] for n in (1, 2, 4, 8)))
