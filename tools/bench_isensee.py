#!/usr/bin/env python3
"""Training step of the Isensee 3-D model at the reference defaults (1x128x128x128, depth 5, 16 base filters), bf16,
through the layer-graph engine.  Prints ms/step and, with --profile, the per-op HIP-event times."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from fetal_net.metrics import dice_coefficient_loss
    from fetal_net.model import isensee2017_model_3d
    shape = (1, 128, 128, 128)
    model = isensee2017_model_3d(input_shape=shape, loss_function=dice_coefficient_loss)
    # live data (round 4): two batches of the learnable task, >= 1.5 s of untimed steps (profiles/r04_data_dependence.json)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    pool = [LT.device_batch(k * a.batch, a.batch, shape[1:]) for k in range(2)]
    t0, i = time.time(), 0
    while time.time() - t0 < 1.5:
        model.train_on_batch(*pool[i % 2])
        i += 1
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(a.steps):
        model.train_on_batch(*pool[(i + k) % 2])
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    print(json.dumps({"workload": "isensee2017_model_3d defaults, batch %d" % a.batch, "ms_per_step": dt * 1e3, "patches_per_s": a.batch / dt}))


if __name__ == "__main__":
    main()
