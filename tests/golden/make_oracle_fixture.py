#!/usr/bin/env python3
"""Pins the CPU oracle itself ("oracle logits for cfg1 on seeded weights/volume"): BASELINE configs[0] - depth 3, 8 base
filters, one 1x16x64x64 patch - with the SURVEY 8d recipe (weights seed 42, volume seed 1234, labels seed 1235), forward in float64.
    python tests/golden/make_oracle_fixture.py      -> tests/golden/oracle_cfg1_golden.npz
The arithmetic is cross-checked at test time against the slow numpy chain (tests/test_oracle_unet.py); this file only guards the oracle
against silent drift between rounds (the GPU parity tests compare against the LIVE oracle)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import unet_oracle as O


def compute():
    shape = (1, 1, 16, 64, 64)
    spec = O.Spec(shape[1:], depth=3, n_base_filters=8)
    W = spec.init_weights(42)
    x, y = O.synthetic_batch(shape)
    logits, probs = O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(x, dtype=torch.float64))
    dice = O.dice_coefficient_t(torch.tensor(y, dtype=torch.float64), probs)
    return dict(logits=logits.detach().numpy().astype(np.float32), dice=np.float64(dice), x_sum=np.float64(x.sum()), y_sum=np.int64(y.sum()),
                w_first=W[sorted(W)[0]].astype(np.float32).ravel()[:16])


if __name__ == "__main__":
    out = compute()
    np.savez_compressed(os.path.join(HERE, "oracle_cfg1_golden.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})
