#!/usr/bin/env python3
"""End-to-end training throughput THROUGH the reference API (`train_model` -> `fit_generator`) at BASELINE config 2, fed (a) by a host
generator that yields float64 numpy batches like the reference's (one ready batch re-yielded: isolates the boundary cost: conversion,
PCIe upload, per-batch metric read-back) and (b) by the device generator.  bench.py's headline has the batch resident in HBM."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import numpy as np
import torch


def main():
    from bench_sampler import FULL, Vols
    from fetal_net.device_generator import DeviceDataFile, device_data_generator
    from fetal_net.metrics import dice_coefficient_loss
    from fetal_net.model import unet_model_3d
    patch, B, steps = (64, 128, 128), 4, int(os.environ.get("FMRI_BENCH_FIT_STEPS", "60"))
    model = unet_model_3d(input_shape=(1,) + patch, depth=4, n_base_filters=32, initial_learning_rate=1e-4, loss_function=dice_coefficient_loss)
    # one batch of the learnable task (float64 on the host, as the reference's generator yields): a live network, as in bench.py
    import learnable_task as LT
    xd_, yd_ = LT.device_batch(LT.HELD_OUT + 700_000, B, patch)
    xb, yb = xd_.cpu().numpy().astype(np.float64), yd_.cpu().numpy()

    def host_gen():
        while True:
            yield xb, yb

    ddf = DeviceDataFile(Vols(6, (96, 192, 192)), patch)
    dev_gen = device_data_generator(ddf, list(range(6)), batch_size=B, patch_shape=patch, augment=FULL, truth_index=0, truth_size=patch[2], is3d=True,
                                    categorical=False, skip_blank=False)
    # the rate the device generator is held against: batches of the SAME generator made beforehand and cycled from HBM (one batch re-yielded
    # would be learnt by heart within the leg: gradients near zero, a higher clock - profiles/r04_data_dependence.json)
    pool_gen = device_data_generator(ddf, list(range(6)), batch_size=B, patch_shape=patch, augment=FULL, truth_index=0, truth_size=patch[2], is3d=True,
                                     categorical=False, skip_blank=False, noise_seed=1)
    pool = [next(pool_gen) for _ in range(32)]
    pool_gen.close()

    def resident_gen():
        while True:
            for xr, yr in pool:
                yield xr, yr

    out = {"augment": "reference default (fetal/config_utils.py:81-123)"}
    only = os.environ.get("FMRI_BENCH_FIT_ONLY", "")          # "host" | "device": one leg only (tools/trace_fit.sh)

    def leg(g, n):
        torch.cuda.synchronize()
        t0 = time.time()
        model.fit_generator(g, steps_per_epoch=n, epochs=1, verbose=0)
        torch.cuda.synchronize()
        return n * B / (time.time() - t0)

    gens = {"host_float64_generator": host_gen(), "resident_device_batch": resident_gen(), "device_generator": dev_gen}
    if only:
        gens = {k: g for k, g in gens.items() if k.startswith(only)}
    for g in gens.values():
        model.fit_generator(g, steps_per_epoch=5, epochs=1, verbose=0)          # warm-up
    # the legs interleaved, three rounds: a box drifts by ~1 % over a minute (it warms), more than the difference looked for
    rates = {k: [] for k in gens}
    for _ in range(3):
        for k, g in gens.items():
            rates[k].append(leg(g, steps))
    for k, v in rates.items():
        out[k + "_patches_per_s"] = sum(v) / len(v)
        out[k + "_legs"] = v
    if "device_generator" in rates and "resident_device_batch" in rates:
        out["generator_vs_resident"] = out["device_generator_patches_per_s"] / out["resident_device_batch_patches_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
