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
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, *spatial, 1).astype(np.float32)).cuda().to(torch.bfloat16)
y = torch.from_numpy((rs.rand(B * int(np.prod(spatial))) > 0.7).astype(np.uint8)).cuda()
out = {}
for name, kw in (("plain", {}), ("batch_norm", dict(norm="batch")), ("instance_norm", dict(norm="instance")), ("deconvolution", dict(deconvolution=True))):
    eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32, **kw), B, dtype=torch.bfloat16)
    for _ in range(3):
        eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(8):
        eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    out[name] = round((time.time() - t0) / 8 * 1e3, 2)
    del eng
    torch.cuda.empty_cache()
print(json.dumps({"ms_per_step": out}))
