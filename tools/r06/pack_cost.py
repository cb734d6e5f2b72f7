#!/usr/bin/env python3
"""What the weight-image repack costs the configs[1] step: the same engine timed with the repack as shipped (side stream, under the next
step's first convolutions), with it skipped (the images go stale: a timing-only arm) and with the optimizer skipped as well.  Interleaved
reps on one box.  usage: pack_cost.py [reps] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import learnable_task as LT
from fmri_hip.engine import UNetEngine, UNetPlan

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, spatial = 4, (64, 128, 128)
eng = UNetEngine(UNetPlan(1, spatial, depth=4, n_base_filters=32), N, dtype=torch.bfloat16)
x, y = LT.device_batch(LT.HELD_OUT + 900_000, N, spatial)
x, y = x.to(torch.bfloat16).reshape(N, *spatial, 1).contiguous(), y.reshape(-1).contiguous()
real_refresh, real_adam = eng.refresh_weight_copies, eng.adam_step


def run(arm):
    eng.refresh_weight_copies = real_refresh if arm == "shipped" else (lambda overlap=False: None)
    eng.adam_step = (lambda lr, **kw: None) if arm == "no_optimizer" else real_adam
    for _ in range(5):
        eng.train_step(x, y, 1e-5)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(K):
        eng.train_step(x, y, 1e-5)
    torch.cuda.synchronize()
    return (time.time() - t0) / K * 1e3


arms = ["shipped", "no_repack", "no_optimizer"]
res = {a: [] for a in arms}
for r in range(REPS):
    for a in arms:
        res[a].append(run(a))
for a in arms:
    v = sorted(res[a])
    print("%-13s median %.3f ms  (min %.3f max %.3f)  %s" % (a, v[len(v) // 2], v[0], v[-1], " ".join("%.3f" % t for t in res[a])))
