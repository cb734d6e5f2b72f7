#!/usr/bin/env python3
"""K one-stream training steps of configs[3] (2-D U-Net, 64 x 256x256x5, depth 4 / 32 filters, bf16) and nothing else: the program the
counter passes of tools/r06/collect_cfg3.sh profile (exclusive launches; after one untimed step).  usage: step_2d.py [K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip.engine import UNetEngine, UNetPlan

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, X, Y, C = 64, 256, 256, 5
eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), B, dtype=torch.bfloat16)
eng._wg_stream = None
g = torch.Generator().manual_seed(0)
x = torch.randn((1, B, X, Y, C), generator=g).cuda().to(torch.bfloat16)
y = (torch.rand((B * X * Y,), generator=g) > 0.7).to(torch.uint8).cuda()
for _ in range(K):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
