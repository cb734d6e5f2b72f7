#!/usr/bin/env python3
"""Generates tests/golden/val_dice_oracle.json: the CPU oracle (oracle/unet_oracle.py, fp32, Keras-Adam) trained on the learnable
synthetic task of tools/learnable_task.py with exactly the schedule tests/test_gpu_val_dice.py drives through train_model():

    depth 4 / 32 filters (every level on the MFMA kernels at this size), 1 x 32x64x128 patches, lr 1e-4, EPOCHS x STEPS Adam steps on
    training seeds 0, 1, 2, ...; after every epoch the soft Dice (reference metrics.py:11-15) of VAL held-out batches (Keras: val_loss =
    -mean of the per-batch values); at the end soft and hard Dice (reference fetal/evaluate.py:16-17, p > 0.5) over the held-out batches
    and the hard Dice of a 48x96x192 held-out volume reconstructed by the oracle tiler (oracle/tiler_oracle.py, overlap 0.5).

The file holds numbers only (no weights); the GPU test compares the bf16 and the fp32 engine with them.  FMRI_LIVE_ORACLE=1 makes the test
run this very function on the GPU box instead of reading the file.  ~8 minutes on 8 cores.

    python tests/golden/make_val_dice_fixture.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

SPATIAL, BATCH, EPOCHS, STEPS, VAL, LR = (32, 64, 128), 1, 4, 25, 4, 1e-4
VOLUME, OVERLAP, SEED_W = (48, 96, 192), 0.5, 42
VOLUME_SEED_OFFSET = 500_000


def run_oracle(threads=None, log=None):
    from oracle import tiler_oracle, unet_oracle as O
    import learnable_task as LT
    torch.set_num_threads(threads or min(32, os.cpu_count() or 1))
    spec = O.Spec((1,) + SPATIAL, depth=4, n_base_filters=32)
    W = spec.init_weights(SEED_W)
    opt = O.KerasAdam(W, lr=LR, dtype=np.float32)
    held = [LT.host_batch(LT.HELD_OUT + k * BATCH, BATCH, SPATIAL, np.float32) for k in range(VAL)]

    def probs_of(x):
        with torch.no_grad():
            return O.forward(spec, O.to_torch(W, torch.float32), torch.tensor(np.asarray(x, np.float32)))[1].numpy()

    out = dict(config=dict(spatial=SPATIAL, batch=BATCH, epochs=EPOCHS, steps_per_epoch=STEPS, validation_steps=VAL, lr=LR, seed_weights=SEED_W,
                           volume=VOLUME, overlap=OVERLAP, depth=4, n_base_filters=32), train_loss=[], val_soft_dice_per_epoch=[])
    k = 0
    t0 = time.time()
    for ep in range(EPOCHS):
        for _ in range(STEPS):
            x, y = LT.host_batch(k * BATCH, BATCH, SPATIAL, np.float32)
            out["train_loss"].append(O.train_step(spec, W, opt, x, y, dtype=torch.float32)["loss"])
            k += 1
        out["val_soft_dice_per_epoch"].append(float(np.mean([LT.soft_dice(y, probs_of(x)) for x, y in held])))
        if log:
            log("oracle epoch %d: train loss %.4f, held-out soft Dice %.4f (%.0f s)" % (ep + 1, np.mean(out["train_loss"][-STEPS:]),
                                                                                        out["val_soft_dice_per_epoch"][-1], time.time() - t0))
    P = [probs_of(x) for x, _ in held]
    out["held_out_soft_dice"] = float(np.mean([LT.soft_dice(y, p) for (_, y), p in zip(held, P)]))
    out["held_out_hard_dice"] = float(np.mean([LT.hard_dice(y, p > 0.5) for (_, y), p in zip(held, P)]))

    class OracleModel:
        output_shape = (None, 1) + SPATIAL

        def predict(self, xb):
            return probs_of(xb)

    vx, vy = LT.host_patch(LT.HELD_OUT + VOLUME_SEED_OFFSET, VOLUME)
    rec = tiler_oracle.patch_wise_prediction(OracleModel(), vx[None].astype(np.float64), SPATIAL, overlap_factor=OVERLAP)
    out["volume_hard_dice"] = LT.hard_dice(vy, rec[..., 0] > 0.5)
    out["volume_soft_dice"] = LT.soft_dice(vy, rec[..., 0])
    out["torch_threads"] = torch.get_num_threads()
    out["seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    res = run_oracle(log=lambda s: print(s, flush=True))
    path = os.path.join(ROOT, "tests", "golden", "val_dice_oracle.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path, {k: v for k, v in res.items() if k not in ("train_loss", "config")})
