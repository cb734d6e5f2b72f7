#!/usr/bin/env python3
"""The planar conv blocks of configs[3] that carry a tail (fmri_conv3d_fwd_tail_planar): the fused launch against conv + separate
MaxPooling2D / final 1x1 conv kernels, per layer.  usage: bench_ptail.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops

IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bf = torch.bfloat16
CASES = [("enc0b pool", 64, 256, 256, 32, 64, "pool"), ("enc1b pool", 64, 128, 128, 64, 128, "pool"), ("enc2b pool", 64, 64, 64, 128, 256, "pool"),
         ("dec0b logits", 64, 256, 256, 64, 64, "logits")]


def timed(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(IT):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / IT * 1e3


for name, S, H, W, C0, Cout, kind in CASES:
    x = torch.randn((1, S, H, W, C0), device="cuda").to(bf)
    w = (torch.randn((27, Cout, C0), device="cuda") * 0.05).to(bf)
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty((1, S, H, W, Cout), device="cuda", dtype=bf)
    pool = torch.empty((1, S, H // 2, W // 2, Cout), device="cuda", dtype=bf)
    w1 = torch.randn(Cout, device="cuda")
    b1 = torch.zeros(1, device="cuda")
    lg = torch.empty(S * H * W, device="cuda")
    lg2 = torch.empty((S * H * W, 1), device="cuda")
    t_conv = timed(lambda: ops.conv3d_fwd(x, None, w, b, y, planar=True))
    if kind == "pool":
        t_sep = timed(lambda: ops.maxpool_fwd(y, pool, planar=True))
        t_fused = timed(lambda: ops.conv3d_fwd_tail(x, w, b, y, pool=pool, planar=True))
    else:
        t_sep = timed(lambda: ops.conv1x1_fwd(y, w1.reshape(1, Cout), b1, lg2))
        t_fused = timed(lambda: ops.conv3d_fwd_tail(x, w, b, y, w1=w1, b1=b1, logits=lg, planar=True))
    print("%-14s conv %7.1f us  + separate %6.1f = %7.1f us   fused %7.1f us  (tail costs %+.1f us, saves %.1f)" %
          (name, t_conv, t_sep, t_conv + t_sep, t_fused, t_fused - t_conv, t_conv + t_sep - t_fused))
