#!/usr/bin/env python3
"""Exclusive (one-stream) time of every op of the configs[3] 2-D training step (64 x 256x256x5, depth 4 / 32 filters, bf16), with the
algorithmic MFMA fraction of the 3x3 convs (2*9*Cin*Cout*pixels / time / 2.5 PF).  Tuning aid: python tools/per_layer_2d.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops
from fmri_hip.engine import UNetEngine, UNetPlan

rec = {}
on = [False]


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        if not on[0]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        shp = tuple(tuple(t.shape) for t in a[:6] if torch.is_tensor(t))
        rec.setdefault((name, shp), []).append((e0, e1))
        return r
    setattr(ops, name, f)


for n in ("conv3d_fwd", "conv3d_fwd_tail", "conv3d_dgrad", "conv3d_wgrad", "conv3d_upcat_fwd", "conv3d_upcat_dgrad", "conv3d_upcat_wgrad", "maxpool_fwd",
          "maxpool_bwd", "upsample_bwd", "conv1x1_fwd", "conv1x1_bwd", "sigmoid_dice_fwd", "sigmoid_loss_bwd", "adam_step", "pack_weights",
          "conv3d_pack_up_weights"):
    wrap(n)
B, X, Y, C = 64, 256, 256, 5
eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), B, dtype=torch.bfloat16)
eng._wg_stream = None
g = torch.Generator().manual_seed(0)
x = torch.randn((1, B, X, Y, C), generator=g).cuda().to(torch.bfloat16)
y = (torch.rand((B * X * Y,), generator=g) > 0.7).to(torch.uint8).cuda()
for _ in range(3):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
on[0] = True
K = 5
for _ in range(K):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
rows = []
for (name, shp), ev in rec.items():
    ms = sum(a.elapsed_time(b) for a, b in ev) / K
    rows.append((ms, name, shp, len(ev) // K))
tot = sum(r[0] for r in rows)
for ms, name, shp, n in sorted(rows, reverse=True):
    fl = ""
    if name.startswith("conv3d") and "pack" not in name:
        # first tensor = input (or dy), find Cin / Cout from the shapes
        try:
            pix = shp[0][1] * shp[0][2] * shp[0][3] if "upcat" not in name else None
        except Exception:
            pix = None
    print("%8.3f ms  x%d  %-22s %s" % (ms, n, name, shp))
print("total %.3f ms per step (exclusive)" % tot)
