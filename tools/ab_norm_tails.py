#!/usr/bin/env python3
"""Interleaved A/B of the normalisation tails inside the full training step of a normalised variant (one engine, the switch read per call):
   ab_norm_tails.py batch_norm|instance_norm [reps]   ->  ms per step for FMRI_NORM_FUSE = 0, 1, 2, 3 and FMRI_NORM_FUSE_MAXLEVEL = 0, 1"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch
from fmri_hip.engine import UNetEngine, UNetPlan

kw = {"batch_norm": dict(norm="batch"), "instance_norm": dict(norm="instance")}[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
spatial, B = (64, 128, 128), 4
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, *spatial, 1).astype(np.float32)).cuda().to(torch.bfloat16)
y = torch.from_numpy((rs.rand(B * int(np.prod(spatial))) > 0.7).astype(np.uint8)).cuda()
eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32, **kw), B, dtype=torch.bfloat16)
CONFIGS = [("0", "99"), ("1", "99"), ("2", "99"), ("3", "99"), ("3", "0"), ("3", "1"), ("1", "0"), ("1", "1")]
for _ in range(30):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
res = {c: [] for c in CONFIGS}
for r in range(reps):
    for c in CONFIGS:
        os.environ["FMRI_NORM_FUSE"], os.environ["FMRI_NORM_FUSE_MAXLEVEL"] = c
        for _ in range(3):
            eng.train_step(x, y, 1e-4)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(12):
            eng.train_step(x, y, 1e-4)
        torch.cuda.synchronize()
        res[c].append((time.time() - t0) / 12 * 1e3)
for c in CONFIGS:
    print("%s FUSE=%s MAXLEVEL=%-2s: %s  mean %.2f ms" % (sys.argv[1], c[0], c[1], " ".join("%.2f" % t for t in res[c]), sum(res[c]) / len(res[c])))
