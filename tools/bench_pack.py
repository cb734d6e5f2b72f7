#!/usr/bin/env python3
"""Weight-image packing per layer of BASELINE configs[1] (tuning aid): fp32 master -> bf16 forward + tap-flipped transposed image
(plain layers) / the 8-parity pre-summed filters (decoder 'a' layers).  Prints us per call and the effective GB/s of read + written bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops

bf = torch.bfloat16
PLAIN = [(32, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (256, 256), (128, 128), (64, 64)]
UP = [(512, 256, 256), (256, 128, 128), (128, 64, 64)]


def timeit(f, n=20):
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


tot = 0.0
for Cin, Cout in PLAIN:
    w = torch.randn(27, Cout, Cin, device="cuda")
    wf = torch.empty(27, Cout, Cin, device="cuda", dtype=bf)
    wd = torch.empty(27, Cin, Cout, device="cuda", dtype=bf)
    us = timeit(lambda: ops.pack_weights(w, wf, wd))
    tot += us
    print("plain %4d -> %4d : %7.1f us  %6.0f GB/s" % (Cin, Cout, us, w.numel() * 8 / us / 1e3))
for C0, C1, Cout in UP:
    w = torch.randn(27, Cout, C0 + C1, device="cuda")
    up_f = torch.empty(8, 8, Cout, C0, device="cuda", dtype=bf)
    up_d = torch.empty(8, 8, C0, Cout, device="cuda", dtype=bf)
    sk_f = torch.empty(27, Cout, C1, device="cuda", dtype=bf)
    sk_d = torch.empty(27, C1, Cout, device="cuda", dtype=bf)
    us = timeit(lambda: ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d))
    tot += us
    print("up    %4d+%3d -> %4d : %7.1f us  %6.0f GB/s" % (C0, C1, Cout, us, (w.numel() * 4 + (up_f.numel() + sk_f.numel()) * 4) / us / 1e3))
print("total %.1f us per step" % tot)
