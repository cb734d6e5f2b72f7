#!/usr/bin/env python3
"""Normalisation tails of the conv launches against the separate passes they replace (tuning aid), BASELINE configs[1] layer shapes:
   conv3d_fwd  vs conv3d_fwd_stats            (+ the reduction pass of norm_act_fwd that the tail saves)
   conv3d_dgrad vs conv3d_dgrad_norm          (+ the reduction pass of norm_act_bwd that the tail saves)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops

bf = torch.bfloat16
N = 4
SHAPES = [(64, 128, 128, 32, 64), (64, 128, 128, 64, 64), (64, 128, 128, 64, 32), (32, 64, 64, 64, 128), (32, 64, 64, 128, 128), (16, 32, 32, 128, 256),
          (16, 32, 32, 256, 256)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for per in (0, 1):
    print("per_instance = %d" % per)
    for D, H, W, Cin, Cout in SHAPES:
        if not ops.conv3d_fwd_ntail_ok(Cin, 0, Cout, N, D, H, W, bf):
            print("%3dx%3dx%3d %3d->%3d: no tail (launch too small)" % (D, H, W, Cin, Cout))
            continue
        G = N if per else 1
        x = torch.randn((N, D, H, W, Cin), device="cuda").to(bf)
        w = (torch.randn((27, Cout, Cin), device="cuda") * 0.05).to(bf)
        b = torch.zeros(Cout, device="cuda")
        y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=bf)
        ws = torch.zeros(ops.norm_tail_ws_doubles(G, max(Cin, Cout)), dtype=torch.float64, device="cuda")
        stats = torch.zeros((G, Cout, 3), device="cuda")
        gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
        t0 = timeit(lambda: ops.conv3d_fwd(x, None, w, b, y, act=0))
        t1 = timeit(lambda: ops.conv3d_fwd_stats(x, None, w, b, y, ws, per, act=0))
        y2 = torch.empty_like(y)
        ta = timeit(lambda: ops.norm_act_fwd(y, gamma, beta, y2, stats, ws, per, act=1))
        tb = timeit(lambda: ops.norm_act_fwd_pre(y, gamma, beta, y2, stats, ws, per, act=1))
        # input gradient of this conv (dy has Cout channels, the result Cin) into a normalised block of Cin channels
        ok5 = ops.conv3d_fwd_ntail_ok(Cout, 0, Cin, N, D, H, W, bf)
        line = "%3dx%3dx%3d %3d->%3d: fwd %7.1f | +stats %7.1f us (%+5.1f%%), reduction pass saved %6.1f us" % (D, H, W, Cin, Cout, t0, t1, (t1 / t0 - 1) * 100, ta - tb)
        if ok5:
            wd = (torch.randn((27, Cin, Cout), device="cuda") * 0.05).to(bf)
            dz = torch.empty_like(x)
            st5 = torch.zeros((G, Cin, 3), device="cuda")
            st5[..., 1] = 1.0
            nss = torch.zeros((G, Cin, 2), device="cuda")
            g5, b5 = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
            ops.norm_scale_shift(st5, g5, b5, nss)
            ws5 = ws
            t2 = timeit(lambda: ops.conv3d_dgrad(y, wd, dz))
            t3 = timeit(lambda: ops.conv3d_dgrad_norm(y, wd, x, nss, dz, ws5, per, act=1))
            dx, dg, db = torch.empty_like(x), torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
            tc = timeit(lambda: ops.norm_act_bwd(x, None, dz, g5, st5, dx, dg, db, ws5, per, act=1, beta=b5))
            td = timeit(lambda: ops.norm_act_bwd_pre(x, dz, g5, st5, dx, dg, db, ws5, per))
            line += " || dgrad %7.1f | +norm %7.1f us (%+5.1f%%), passes saved %6.1f us" % (t2, t3, (t3 / t2 - 1) * 100, tc - td)
        print(line)
