#!/usr/bin/env python3
"""BASELINE configs[3]: 2-D mode, 256x256 slices with 5 input channels, batch 64 per GPU, depth 4 / 32 filters, bf16: full training step
(fwd + Dice + bwd + Adam) in slices/s on one GPU - the `secondary.cfg3` leg of bench.py (live learnable data since round 4).  Prints JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch

if __name__ == "__main__":
    import bench
    torch.cuda.set_device(0)
    print(json.dumps(bench.cfg3_leg(steps=20, warmup=5)))
