#!/usr/bin/env python3
"""The reference's training flow end to end on this package: fetal_net.generator.get_training_and_validation_generators (device generator,
the reference's DEFAULT augmentation on the training split, none on the validation split) -> fetal_net.training.train_model, depth-4 / 32-filter
3-D U-Net on 64x128x128 patches of learnable-task volumes (tools/learnable_task.py).  Prints one JSON line: held-out soft Dice per epoch and the
training rate.  Does the network LEARN through the augmenting generator, at what speed?"""
import json
import os
import random
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import learnable_task as LT
from bench_sampler import FULL
from fetal_net.generator import get_training_and_validation_generators
from fetal_net.metrics import dice_coefficient_loss
from fetal_net.model import unet_model_3d
from fetal_net.training import train_model


class _R:
    pass


def main():
    patch, B = (64, 128, 128), 4
    epochs, steps, vsteps = int(os.environ.get("EPOCHS", "6")), int(os.environ.get("STEPS", "100")), 8
    f = _R()
    f.root = _R()
    pairs = [LT.device_patch(LT.HELD_OUT + 600_000 + k, (96, 192, 192)) for k in range(10)]
    f.root.data = [x.float().cpu().numpy() for x, _ in pairs]
    f.root.truth = [y.cpu().numpy().astype(np.uint8) for _, y in pairs]
    f.root.subject_ids = [("vol%d" % k).encode() for k in range(10)]
    random.seed(0)
    np.random.seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        tr, va, _, _ = get_training_and_validation_generators(f, B, 1, os.path.join(tmp, "tr.pkl"), os.path.join(tmp, "va.pkl"), os.path.join(tmp, "te.pkl"),
                                                              patch_shape=patch, data_split=0.8, augment=FULL, validation_batch_size=B, truth_index=0,
                                                              truth_size=patch[2], patches_per_epoch=steps * B, categorical=False, is3d=True,
                                                              skip_blank_train=True, verbose=False)
        model = unet_model_3d(input_shape=(1,) + patch, depth=4, n_base_filters=32, initial_learning_rate=1e-4, loss_function=dice_coefficient_loss)
        torch.cuda.synchronize()
        t0 = time.time()
        hist = train_model(model, os.path.join(tmp, "model"), tr, va, steps_per_epoch=steps, validation_steps=vsteps, initial_learning_rate=1e-4,
                           n_epochs=epochs, output_folder=tmp)
        torch.cuda.synchronize()
        dt = time.time() - t0
        h = hist.history if hasattr(hist, "history") else hist
        print(json.dumps({"workload": "get_training_and_validation_generators (reference default augmentation) -> train_model, 64x128x128 x 4, bf16",
                          "epochs": epochs, "steps_per_epoch": steps, "validation_steps": vsteps,
                          "val_soft_dice_per_epoch": [-float(v) for v in h["val_loss"]], "train_soft_dice_per_epoch": [-float(v) for v in h["loss"]],
                          "seconds": round(dt, 2), "patches_per_s_including_validation": round(epochs * (steps + vsteps) * B / dt, 1)}))


if __name__ == "__main__":
    main()
