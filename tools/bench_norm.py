#!/usr/bin/env python3
"""Effective HBM rate of the normalisation passes (fmri_norm_act_fwd / _bwd) on the tensors of the benchmark's level 0 and of the Isensee
defaults: bytes = the tensors each pass has to touch once (fwd: x twice + y; bwd: x, y, dy for the sums, x, y, dy + dx for the apply)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops

for name, N, V, C, inst in (("unet level 0, instance", 4, 64 * 128 * 128, 64, 1), ("unet level 0, batch", 4, 64 * 128 * 128, 64, 0),
                            ("unet level 1, instance", 4, 32 * 64 * 64, 128, 1), ("isensee level 0 (16 -> 32 padded)", 1, 128 ** 3, 32, 1),
                            ("discriminator level 0", 8, 32 * 64 * 128, 32, 1)):
    x = torch.randn((N, V, C), device="cuda").to(torch.bfloat16)
    y, dy, dx = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    stats = torch.zeros((N if inst else 1, C, 3), device="cuda")
    ws = torch.zeros((N, C, 2), dtype=torch.float64, device="cuda")
    nbytes = x.numel() * 2

    def t(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    tf = t(lambda: ops.norm_act_fwd(x, gamma, beta, y, stats, ws, inst, eps=1e-3, eps_on_std=bool(inst), act=2, alpha=0.3))
    tb = t(lambda: ops.norm_act_bwd(x, y, dy, gamma, stats, dx, dg, db, ws, inst, act=2, alpha=0.3))
    tx = t(lambda: ops.norm_act_bwd(x, None, dy, gamma, stats, dx, dg, db, ws, inst, act=2, alpha=0.3, beta=beta))
    print("%-36s %6.1f MB  fwd %.3f ms = %.2f TB/s (3 tensor passes)   bwd %.3f ms = %.2f TB/s (7 tensor passes)   bwd without y %.3f ms = %.2f TB/s (5 passes)" % (
        name, nbytes / 1e6, tf * 1e3, 3 * nbytes / tf / 1e12, tb * 1e3, 7 * nbytes / tb / 1e12, tx * 1e3, 5 * nbytes / tx / 1e12))
