#!/usr/bin/env python3
"""BASELINE configs[4]: sliding-window inference over one 160x256x256 volume, patch 64x128x128, overlap_factor 0.5 (36 tiles),
through fetal_net.prediction.patch_wise_prediction (device gather -> network -> float64 overlap-add, hipGraph).  Prints JSON."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch

import fetal_net.model as fmodel
from fetal_net.prediction import patch_wise_prediction

if __name__ == "__main__":
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    patch = (64, 128, 128)
    model = fmodel.unet_model_3d(input_shape=(1,) + patch)
    data = np.random.RandomState(0).randn(1, 160, 256, 256).astype(np.float32)
    out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5, batch_size=bs)   # builds graphs
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5, batch_size=bs)
        times.append(time.perf_counter() - t0)
    # device-only part: replay the captured graphs again
    st = model._tile_state
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for B, pb in st["per_b"].items():
        reps = sum(1 for i in range(0, 36, max(st["per_b"])) if min(max(st["per_b"]), 36 - i) == B)     # groups of this size in the volume
        for _ in range(reps):
            pb["graph"].replay() if pb["graph"] is not None else pb["body"]()
    torch.cuda.synchronize()
    dev = time.perf_counter() - t0
    print(json.dumps({"workload": "configs[4]: 160x256x256 volume, patch 64x128x128, overlap_factor 0.5, 36 tiles, caller batch %d, device groups %s" % (bs, sorted(st["per_b"])),
                      "end_to_end_s_per_volume": min(times), "device_tile_loop_s": dev, "out_shape": list(out.shape),
                      "tiles_per_s_device": 36 / dev, "fwd_tflops_device": 36 * 1893.5e9 / dev / 1e12,
                      "finite": bool(np.isfinite(out).all())}))
