#!/usr/bin/env python3
"""Full training step of the config-2 U-Net (depth 4 / 32 filters, 4 x 64x128x128, bf16) in its reference variants
(reference unet3d/unet.py:17-20: batch_normalization, instance normalisation, deconvolution): ms per step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch
from fmri_hip.engine import UNetEngine, UNetPlan

spatial, B = (64, 128, 128), 4
# live data (round 4): two batches of the learnable task, >= 1.5 s of untimed steps - with labels independent of the image the net collapses to
# all-foreground, the backward kernels multiply ~0 gradients and the power-limited clock rises (profiles/r04_data_dependence.json)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import learnable_task as LT
POOL = []
for k in range(2):
    xb, yb = LT.device_batch(k * B, B, spatial)
    POOL.append((xb.to(torch.bfloat16).reshape(B, *spatial, 1).contiguous(), yb.reshape(-1).contiguous()))


def warm(eng, seconds=1.5):
    t0, i = time.time(), 0
    while time.time() - t0 < seconds:
        eng.train_step(*POOL[i % 2], 1e-4)
        i += 1
        if i % 5 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return i
out = {}
for name, kw in (("plain", {}), ("batch_norm", dict(norm="batch")), ("instance_norm", dict(norm="instance")), ("deconvolution", dict(deconvolution=True))):
    eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32, **kw), B, dtype=torch.bfloat16)
    i0 = warm(eng)
    t0 = time.time()
    for i in range(20):
        eng.train_step(*POOL[(i0 + i) % 2], 1e-4)
    torch.cuda.synchronize()
    out[name] = round((time.time() - t0) / 20 * 1e3, 2)
    del eng
    torch.cuda.empty_cache()
print(json.dumps({"ms_per_step": out}))
