#!/usr/bin/env python3
"""A few training steps of one reference variant of the config-2 U-Net (for rocprofv3):  step_variant.py batch_norm|instance_norm|deconvolution|plain [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch
from fmri_hip.engine import UNetEngine, UNetPlan

kw = {"plain": {}, "batch_norm": dict(norm="batch"), "instance_norm": dict(norm="instance"), "deconvolution": dict(deconvolution=True)}[sys.argv[1]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spatial, B = (64, 128, 128), 4
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, *spatial, 1).astype(np.float32)).cuda().to(torch.bfloat16)
y = torch.from_numpy((rs.rand(B * int(np.prod(spatial))) > 0.7).astype(np.uint8)).cuda()
eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32, **kw), B, dtype=torch.bfloat16)
if os.environ.get("ONE_STREAM", "0") == "1":
    eng._wg_stream = None
for _ in range(3):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
print("%s: %.2f ms per step" % (sys.argv[1], (time.time() - t0) / steps * 1e3))
