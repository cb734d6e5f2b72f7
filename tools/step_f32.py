#!/usr/bin/env python3
"""K training steps of configs[1] in the fp32 parity mode and nothing else (the program behind profiles/r06_f32_kernel_stats.csv: which
kernels the fp32 mode runs - the fp32 instantiations of the benchmarked MFMA kernels, `k_conv_fwd_ws<1, …, true>` / `k_conv_wgrad_kd<32, false, true>`).
usage: step_f32.py [K] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import learnable_task as LT
from fmri_hip.engine import UNetEngine, UNetPlan

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
spatial = (64, 128, 128)
eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32), N, dtype=torch.float32)
x, y = LT.device_batch(LT.HELD_OUT + 900_000, N, spatial)
x, y = x.float().reshape(N, *spatial, 1).contiguous(), y.reshape(-1).contiguous()
eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(K):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
print("fp32 parity mode: %.1f ms per batch-%d step" % ((time.time() - t0) / K * 1e3, N))
