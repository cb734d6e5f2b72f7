#!/usr/bin/env python3
"""BASELINE configs[3]: 2-D mode, 256x256 slices with 5 input channels, batch 64 per GPU, depth 4 / 32 filters, bf16:
full training step (fwd + Dice + bwd + Adam) in slices/s on one GPU.  Prints JSON."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch
from fmri_hip.engine import UNetEngine, UNetPlan

if __name__ == "__main__":
    B, X, Y, C = 64, 256, 256, 5
    eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), B, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, B, X, Y, C), generator=g).cuda().to(torch.bfloat16)
    y = (torch.rand((B * X * Y,), generator=g) > 0.7).to(torch.uint8).cuda()
    for _ in range(3):
        eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        s = eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    # algorithmic FLOPs of the 3x3 convs, fwd+dgrad+wgrad (9 taps)
    fl = 0.0
    first = eng.plan.enc[0][0]["name"]
    for c in eng.plan.convs_forward_order():
        _, H, W = eng.plan.level_dims(c["level"], B)
        f = 2.0 * 9 * c["cin"] * c["cout"] * B * H * W
        fl += f * (2 if c["name"] == first else 3)
    print(json.dumps({"workload": "configs[3]: 2-D U-Net depth 4 / 32 filters, 64x256x256x5 bf16 per GPU, full training step",
                      "ms_per_step": dt * 1e3, "slices_per_s": B / dt, "conv_tflops": fl / dt / 1e12,
                      "dice": eng.metrics_from_sums(s.cpu().numpy())["dice_coefficient"]}))
