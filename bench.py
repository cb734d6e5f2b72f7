#!/usr/bin/env python3
"""Headline benchmark: 3-D patches/s (64x128x128, bf16) of the full training step (forward + Dice + backward + Adam) of the
depth-4 / 32-base-filter U-Net (BASELINE.json configs[1]; reference fetal_net/model/unet3d/unet.py:17-86 defaults), batch 4
per GPU, synthetic data resident in HBM, random-init (glorot) weights.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) this
process IS one rank; started bare (`python bench.py --gpus 8`) it becomes a parent that starts the N ranks as child processes - before
anything touches the GPU - relays rank 0's JSON line and exits non-zero when any rank fails.  A rank whose WORLD_SIZE differs from
--gpus refuses to run (a single-GPU number can no longer be labelled as a scaling run).

Prints ONE JSON line on rank 0.  Besides the driver contract it carries
  "roofline"     for the dominant kernel (live HIP-event timing of its launches in the timed region: every launch of every 10th timed
                 step - event brackets on every step cost 2.3 % of the step, see --launch-timing-every), and
  "cpu_baseline" the CPU oracle (torch-CPU restatement; Keras/TF are absent here and on the GPU box) timed on the host (the only leg that
                 touches oracle/; its first step's forward pass is also the reference of the next item),
  "parity_mode"  what FMRI_DTYPE=fp32 costs and delivers at this configuration: patches/s of the fp32 step on the fp32 instantiation of the
                 same MFMA kernels, logits / Dice of one patch against that reference (north-star bars 1e-3 / 1e-4).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fetal-mri-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same table
# algorithmic bf16 bytes of the 3x3x3 convs, fwd+dgrad+wgrad, per 64x128x128 patch
ALGO_BYTES_PER_PATCH = 4337e6


def conv_flops(eng):
    """algorithmic FLOPs (2*27*Cin*Cout*voxels) of every 3x3x3 conv launch class in ONE step of this engine"""
    p, N = eng.plan, eng.N
    out = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "fwd_mfma": 0.0, "dgrad_mfma": 0.0, "wgrad_mfma": 0.0, "fwd_dgrad_executed": 0.0}
    upcat = getattr(eng, "upcat", {})
    from fmri_hip._lib import lib, BF16
    first = p.enc[0][0]["name"]
    for c in p.convs_forward_order():
        D, H, W = p.level_dims(c["level"])
        fl = 2.0 * 27 * c["cin"] * c["cout"] * N * D * H * W
        c0, c1 = (c["c_up"], c["c_skip"]) if "c_up" in c else (c["cin"], 0)
        m = lib().fmri_conv3d_uses_mfma(c0, c1, c["cout"], D, H, W, BF16)
        out["fwd"] += fl
        out["wgrad"] += fl
        # MACs the device really executes: the parity form of the decoder 'a' layers does 8 instead of 27 taps on the up-sampled channels
        fl_exec = fl
        if c["name"] in upcat:
            cu, cs = upcat[c["name"]]
            fl_exec = 2.0 * (8 * cu + 27 * cs) * c["cout"] * N * D * H * W
        if m & 1:
            out["fwd_mfma"] += fl
            out["fwd_dgrad_executed"] += fl_exec
        if m & 2:
            out["wgrad_mfma"] += fl
        if c["name"] != first:
            out["dgrad"] += fl
            md = lib().fmri_conv3d_uses_mfma(c["cout"], 0, c["cin"], D, H, W, BF16)
            if md & 1:
                out["dgrad_mfma"] += fl
                out["fwd_dgrad_executed"] += fl_exec
    return out


class LaunchTimer:
    """HIP-event brackets around C-ABI launches on torch's current stream (the stream the kernels run on)."""

    def __init__(self):
        self.rec = {}
        self.detail = {}
        self.on = False

    def wrap(self, ops_mod, fn_name, label_fn, launches=1):
        """`launches`: kernel launches of the labelled kind one call makes (the parity-form ops launch the conv kernel twice)"""
        orig = getattr(ops_mod, fn_name)

        def wrapped(*a, **k):
            if not self.on:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            lab = label_fn(*a, **k)
            fam, detail = lab if isinstance(lab, tuple) else (lab, None)
            self.rec.setdefault(fam, []).append((e0, e1, launches))
            if detail is not None:                      # per-layer view: (pass, C0, C1, Cout, D) identifies a conv of the plan
                self.detail.setdefault(detail, []).append((e0, e1))
            return r

        setattr(ops_mod, fn_name, wrapped)

    def totals_ms(self):
        return {k: (sum(a.elapsed_time(b) for a, b, _ in v), sum(n for _, _, n in v)) for k, v in self.rec.items()}

    def detail_ms(self):
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.detail.items()}


def synthetic_batch(shape, seed_x=1234, seed_y=1235, fg=0.30):
    """SURVEY.md §8(d) recipe: x ~ N(0,1) (the reference z-scores its volumes), y = smooth-noise blob mask with ~30 % foreground."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed_x)
    x = rs.randn(*shape).astype(np.float32)
    rs = np.random.RandomState(seed_y)
    y = np.zeros((shape[0], 1) + tuple(shape[2:]), np.uint8)
    for n in range(shape[0]):
        f = gaussian_filter(rs.randn(*shape[2:]), sigma=[min(4.0, s / 8.0) for s in shape[2:]])
        y[n, 0] = (f > np.quantile(f, 1.0 - fg)).astype(np.uint8)
    return x, y


def cpu_baseline(budget_s=25.0):
    """the oracle's training step on ONE 64x128x128 patch of the same model on the host cores.  torch-CPU/oneDNN does not
    scale to all 256 hardware threads of the GPU box (measured: 16-32 threads are fastest, 256 are 60x slower), so the thread
    count is picked by a 3-point probe on a 1/16-size patch first; `cores` reports the count actually used."""
    from oracle import unet_oracle as O
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, c) for c in (16, 32, 64)})
    spec_s = O.Spec((1, 32, 64, 64))
    Ws = spec_s.init_weights(42)
    xs, ys = O.synthetic_batch((1, 1, 32, 64, 64))
    best, best_dt = cands[0], 1e30
    for c in cands:
        torch.set_num_threads(c)
        opt = O.KerasAdam(Ws, lr=1e-4, dtype=np.float32)
        O.train_step(spec_s, Ws, opt, xs, ys, dtype=torch.float32)
        t0 = time.time()
        O.train_step(spec_s, Ws, opt, xs, ys, dtype=torch.float32)
        dt = time.time() - t0
        if dt < best_dt:
            best, best_dt = c, dt
    torch.set_num_threads(best)
    spec = O.Spec((1, 64, 128, 128))
    W = spec.init_weights(42)
    W0 = dict(W)                            # the initial weights (Adam rebinds the entries of W, it does not write into the arrays)
    x, y = O.synthetic_batch((1, 1, 64, 128, 128))
    opt = O.KerasAdam(W, lr=1e-4, dtype=np.float32)
    t0 = time.time()
    n = 0
    ref = None
    while True:
        r = O.train_step(spec, W, opt, x, y, dtype=torch.float32)
        if n == 0:
            # the first timed step's forward pass, on the initial weights: the reference the parity-mode leg compares the fp32 engine with
            # (the oracle is used in THIS leg only; what leaves it is data)
            ref = {"weights": W0, "x": x, "y": y, "logits": r["logits"], "dice": r["dice"]}
        n += 1
        if n >= 3 or time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "patches/s", "cores": best, "kind": "port",
            "sample": "%d full training steps (fwd+Dice+bwd+Adam, fp32, torch-CPU/oneDNN restatement of the Keras path) on one "
                      "1x64x128x128 patch of the same depth-4/32-filter model, %d threads (best of %s on a probe; host has %d); "
                      "Keras/TF not installed" % (n, best, cands, ncpu)}, ref


def val_dice_leg(batch, steps, spatial=(64, 128, 128), volume=(160, 256, 256), lr=1e-4, log=None):
    """The val-Dice leg of BASELINE.json's metric (VERDICT r3 row g), AFTER the timed region: a fresh bf16 model of the
    benchmarked configuration is trained for `steps` Adam steps on the learnable synthetic task (tools/learnable_task.py: image and label
    from one latent field, batches generated in HBM) THROUGH the reference's API - fetal_net.training.train_model() with its callbacks
    and checkpoints - then
      soft              = -val_loss of the last epoch's Keras log = dice_coefficient (reference metrics.py:11-15) on held-out batches,
      hard_cfg5_volume  = 2|t&p| / (|t| + |p|) (reference fetal/evaluate.py:16-17), p > 0.5, over a held-out 160x256x256 volume
                          reconstructed by fetal_net.prediction.patch_wise_prediction (patch 64x128x128, overlap_factor 0.5, 36 tiles).
    Convergence parity with the fp32 engine and the CPU oracle is what tests/test_gpu_val_dice.py asserts (same task, smaller patches)."""
    import itertools
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    import fetal_net.metrics as FM
    import fetal_net.model as fmodel
    from fetal_net.prediction import patch_wise_prediction
    from fetal_net.training import train_model
    t0 = time.perf_counter()
    model = fmodel.unet_model_3d(input_shape=(1,) + tuple(spatial), depth=4, n_base_filters=32, initial_learning_rate=lr,
                                 loss_function=FM.dice_coefficient_loss)
    epochs = 3 if steps % 3 == 0 else 1
    vsteps = 4
    held = [LT.device_batch(LT.HELD_OUT + k * batch, batch, spatial) for k in range(vsteps)]
    with tempfile.TemporaryDirectory() as tmp:
        t1 = time.perf_counter()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):          # the epoch lines of fit_generator / ModelCheckpoint: not part of the JSON line
            hist = train_model(model, os.path.join(tmp, "fetal_net_model"), LT.device_generator(0, batch, spatial), itertools.cycle(held),
                               steps_per_epoch=steps // epochs, validation_steps=vsteps, initial_learning_rate=lr, n_epochs=epochs,
                               output_folder=tmp).history
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t1
    # the step time ON THIS TASK with the trained weights (resident batch, same loop as the headline's timed region): MI355X's power-limited
    # shader clock depends on the operands' statistics, so the headline's recipe (labels independent of the image - the net ends up
    # predicting the base rate) and a model that is really learning need not run at the same clock
    eng = model.engine(batch)
    xb, yb = LT.device_batch(LT.HELD_OUT + 900_000, batch, spatial)
    xe, ye = model._to_device_x(xb), model._to_device_y(yb)
    for _ in range(10):
        eng.train_step(xe, ye, 0.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(40):
        eng.train_step(xe, ye, 0.0)              # lr 0: the trained weights stay as they are (and the held-out numbers below are unaffected)
    torch.cuda.synchronize()
    task_rate = 40 * batch / (time.perf_counter() - t2)
    eng.t -= 50                                   # (the Adam step counter: these were not training steps)
    # (predict() of a device batch returns a view of the engine's probability buffer: use it before the next call)
    hard_b = float(np.mean([LT.hard_dice(y.cpu().numpy(), (model.predict(x) > 0.5).cpu().numpy()) for x, y in held]))
    vx, vy = LT.device_patch(LT.HELD_OUT + 500_000, tuple(volume))
    rec = patch_wise_prediction(model, vx.cpu().numpy()[None].astype(np.float64), spatial, overlap_factor=0.5)
    return {"soft": -float(hist["val_loss"][-1]), "hard_cfg5_volume": LT.hard_dice(vy.cpu().numpy(), rec[..., 0] > 0.5),
            "soft_cfg5_volume": LT.soft_dice(vy.cpu().numpy(), rec[..., 0]), "hard_held_out_batches": hard_b,
            "patches_per_s_on_this_task": task_rate, "steps": int(steps), "soft_per_epoch": [-float(v) for v in hist["val_loss"]], "train_dice_per_epoch": [-float(v) for v in hist["loss"]],
            "lr": lr, "dtype": "bf16", "held_out_patches": vsteps * batch,
            "task": "tools/learnable_task.py: y = blobs of a smooth latent field (%.0f %% foreground), x = zscore(%.1f * y + N(0,1)); training seeds "
                    "0.., held-out seeds %d.." % (100 * LT.FG, LT.CONTRAST, LT.HELD_OUT),
            "through": "fetal_net.training.train_model -> fit_generator (device batches, %d epochs, callbacks + checkpoints) -> patch_wise_prediction" % epochs,
            "train_model_seconds": round(t_train, 2), "seconds": round(time.perf_counter() - t0, 2)}, model


def parity_mode_leg(batch, ref=None, spatial=(64, 128, 128), steps=3):
    """What the parity mode costs and delivers: FMRI_DTYPE=fp32 at the benchmarked configuration.  fp32 tensors run the
    fp32 instantiation of the SAME kernels (v_mfma_f32_32x32x2_f32: halo box, LDS-DMA, swizzle, asynchronous drain, pooled-copy tail, parity
    form, kd-sharing weight gradient - csrc/conv3d_mfma.hip / conv3d_wgrad.hip, F32), so the north-star's 1e-3 logits bar is met by the
    benchmarked kernel structure.  patches_per_s: full training steps on a live batch; logits_rel / dice_abs: one patch against the CPU oracle
    (reference unet3d/unet.py:68, metrics.py:11-15) on the same weights - `ref` is the first step of the cpu_baseline leg (weights, batch,
    logits, Dice: plain arrays; this leg does not touch oracle/), None with --no-cpu-baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    from fmri_hip.engine import UNetEngine, UNetPlan
    plan = UNetPlan(1, spatial, depth=4, n_base_filters=32)
    eng = UNetEngine(plan, batch, dtype=torch.float32)
    x, y = LT.device_batch(LT.HELD_OUT + 900_000, batch, spatial)
    x, y = x.float().reshape(batch, *spatial, 1).contiguous(), y.reshape(-1).contiguous()
    eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(x, y, 1e-4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = sum(2.0 * 27 * c["cin"] * c["cout"] * batch * int(np.prod(plan.level_dims(c["level"]))) for c in plan.convs_forward_order()) * 3
    del eng
    torch.cuda.empty_cache()
    logits_rel = dice_abs = None
    if ref is not None:
        e1 = UNetEngine(plan, 1, dtype=torch.float32)
        e1.load_keras_weights(ref["weights"])
        e1.forward(torch.from_numpy(ref["x"]).cuda().reshape(1, *spatial, 1).contiguous())
        sums = e1.loss_forward(torch.from_numpy(ref["y"]).cuda().reshape(-1).contiguous())
        torch.cuda.synchronize()
        lg = e1.logits.cpu().numpy().reshape(ref["logits"].shape)
        logits_rel = float(np.abs(lg - ref["logits"]).max() / np.abs(ref["logits"]).max())
        dice_abs = float(abs(e1.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]))
    on_mfma = all(lib_uses_mfma(c, plan) for c in plan.convs_forward_order() if c["cin"] > 1)
    return {"dtype": "fp32", "patches_per_s": batch / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "logits_rel": logits_rel, "logits_bar": 1e-3, "dice_abs": dice_abs, "dice_bar": 1e-4,
            "reference": "the CPU oracle's forward pass of the cpu_baseline leg (same weights, same patch)" if ref is not None else "none: --no-cpu-baseline",
            "f32_mfma_frac": fl / dt / 1e12 / 157.3,
            "kernels": "fp32 instantiation of the benchmarked MFMA kernels (v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s = the fp32 vector rate)" if on_mfma
                       else "VALU kernels (conv3d_generic.hip): FMRI_F32_MFMA=0 or a shape outside the MFMA family",
            "note": "f32_mfma_frac = 3 x 2*27*Cin*Cout*voxels of all 14 convs / step time / 157.3 TF (the reference's tap count; the parity form executes 0.70 of it)"}


def lib_uses_mfma(c, plan):
    from fmri_hip._lib import lib
    D, H, W = plan.level_dims(c["level"])
    return (lib().fmri_conv3d_uses_mfma(c["cin"], 0, c["cout"], D, H, W, 0) & 3) == 3          # dtype 0 = fp32


def reference_api_leg(model, batch, spatial=(64, 128, 128), steps=40, pool=4, generator_leg=True):
    """patches/s of the SAME training step driven through the reference-facing surface (`Model.fit_generator`, what train_model() calls):
    (a) a reference-style host generator yielding float64 numpy batches (generator.py:397-401; a small pool of ready batches of the learnable
    task: isolates the boundary cost), (b) the same batches already in HBM, (c) the bare engine loop on them (= the headline's loop).  All
    three on the same model with the learning rate set to 0 for the duration: the weights - and with them the operands' statistics and the
    power-limited clock - are the same for the three legs (a model that keeps training on four batches drifts towards memorising them, its
    gradients shrink and the later legs would run at a higher clock)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    dev = [LT.device_batch(LT.HELD_OUT + 700_000 + k * batch, batch, spatial) for k in range(pool)]
    host = [(x.cpu().numpy().astype(np.float64), y.cpu().numpy()) for x, y in dev]

    def cycle(items):
        k = 0
        while True:
            yield items[k % len(items)]
            k += 1

    lr_keep = model.optimizer.lr
    model.optimizer.lr = 0.0
    out = {}
    try:
        for name, items in (("host_float64_generator", host), ("device_batches", dev)):
            g = cycle(items)
            model.fit_generator(g, steps_per_epoch=5, epochs=1, verbose=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fit_generator(g, steps_per_epoch=steps, epochs=1, verbose=0)
            torch.cuda.synchronize()
            out[name + "_patches_per_s"] = steps * batch / (time.perf_counter() - t0)
        if generator_leg:
            # (d) fetal_net.device_generator with the reference's DEFAULT augmentation (fetal/config_utils.py:81-123, every entry: affine, contrast,
            # shot / speckle / gaussian noise, elastic transform, coarse dropout) sampling patches from volumes resident in HBM, against (b):
            # what feeding the step from the device sampler costs
            import random
            from fetal_net.device_generator import DeviceDataFile, device_data_generator
            full = {"flip": [0.5, 0.5, 0.5], "permute": False, "translate": (15, 15, 7), "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "poisson_noise": 1,
                    "gaussian_filter": {"prob": 0.0, "max_sigma": 1}, "contrast": {"prob": 0, "min_factor": 0.2, "max_factor": 0.1},
                    "elastic_transform": {"alpha": 5, "sigma": 10}, "coarse_dropout": {"rate": 0.2, "size_percent": [0.10, 0.30], "per_channel": True},
                    "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05}}

            class _Vols(object):
                pass
            vols = _Vols()
            vols.root = _Vols()
            vshape = tuple(s + s // 2 for s in spatial)
            pairs = [LT.device_patch(LT.HELD_OUT + 800_000 + k, vshape) for k in range(3)]
            vols.root.data = [x.float().cpu().numpy() for x, _ in pairs]
            vols.root.truth = [y.cpu().numpy().astype(np.uint8) for _, y in pairs]
            np.random.seed(0)
            random.seed(0)
            ddf = DeviceDataFile(vols, spatial)

            def make(seed):
                return device_data_generator(ddf, [0, 1, 2], batch_size=batch, patch_shape=spatial, augment=full, truth_index=0,
                                             truth_size=spatial[2], is3d=True, categorical=False, skip_blank=False, noise_seed=seed)

            # like against like: the SAME generator's batches made beforehand and cycled from HBM (other data - cleaner, or with other zero
            # fractions - runs the power-limited step at another clock: `device_batches` above is not the reference for this leg)
            pg = make(1)
            ready = [next(pg) for _ in range(8)]
            pg.close()
            g = make(2)

            def timed(gen):
                model.fit_generator(gen, steps_per_epoch=5, epochs=1, verbose=0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.fit_generator(gen, steps_per_epoch=steps, epochs=1, verbose=0)
                torch.cuda.synchronize()
                return steps * batch / (time.perf_counter() - t0)

            r0 = timed(cycle(ready))
            out["device_generator_default_augmentation_patches_per_s"] = timed(g)
            g.close()
            r1 = timed(cycle(ready))
            out["device_generator_ready_batches_patches_per_s"] = [r0, r1]
            out["device_generator_vs_its_ready_batches"] = out["device_generator_default_augmentation_patches_per_s"] / (0.5 * (r0 + r1))
        eng = model.engine(batch)
        res = [(model._to_device_x(x), model._to_device_y(y)) for x, y in dev]
        for k in range(5):
            eng.train_step(*res[k % pool], 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            eng.train_step(*res[k % pool], 0.0)
        torch.cuda.synchronize()
        out["resident_batch_patches_per_s"] = steps * batch / (time.perf_counter() - t0)
    finally:
        model.optimizer.lr = lr_keep
    out["steps"] = steps
    out["note"] = "learning rate 0 during the three legs (same weights, same clock); %d batches of the learnable task" % pool
    return out


def cfg3_leg(steps=10, warmup=3):
    """BASELINE configs[3]: 2-D mode (reference fetal_net/model/unet/unet.py:22-88), 256x256 slices x 5 channels, batch 64 per GPU, depth 4 /
    32 filters, bf16: full training step in slices/s; the 3x3 convs' algorithmic FLOPs (2*9*Cin*Cout*pixels, fwd + dgrad + wgrad) against
    the dense bf16 MFMA peak (11.7 k slices/s is the MFMA ceiling)."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    B, X, Y, C = 64, 256, 256, 5
    eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), B, dtype=torch.bfloat16)
    # live data, as in the headline: a small pool of slice-stack batches of the learnable task, trained on at lr 1e-4
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    pool = []
    for k in range(2):
        xb, yb = LT.device_batch_2d(5_000 + k * B, B, (X, Y), C)
        pool.append((xb.to(torch.bfloat16).unsqueeze(0).contiguous(), yb.reshape(-1).contiguous()))
    t_w = time.perf_counter()
    i = 0
    while i < warmup or time.perf_counter() - t_w < 1.0:          # >= 1 s of untimed steps: a settled clock and a learning network
        eng.train_step(*pool[i % 2], 1e-4)
        i += 1
        if i % 5 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s_ = eng.train_step(*pool[i % 2], 1e-4)
        i += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = 0.0
    first = eng.plan.enc[0][0]["name"]
    for c in eng.plan.convs_forward_order():
        _, H, W = eng.plan.level_dims(c["level"], B)
        fl += 2.0 * 9 * c["cin"] * c["cout"] * B * H * W * (2 if c["name"] == first else 3)
    return {"workload": "configs[3]: 2-D U-Net depth 4 / 32 filters, 64x256x256x5 bf16 per GPU, full training step", "slices_per_s": B / dt,
            "ms_per_step": dt * 1e3, "steps": steps, "conv_tflops_algorithmic": fl / dt / 1e12, "mfma_frac": fl / dt / 1e12 / PEAK_BF16_TFLOPS,
            "train_dice_last_step": eng.metrics_from_sums(s_.cpu().numpy())["dice_coefficient"], "data": "learnable task, 2-D form (tools/learnable_task.py)"}


def cfg4_leg(model=None, reps=3):
    """BASELINE configs[4]: sliding-window inference over one 160x256x256 volume (reference fetal_net/prediction.py:118-210), patch
    64x128x128, overlap_factor 0.5 = the reference's 36 tiles, hipGraph-replayed tile groups: seconds per volume end to end (host volume
    in, float64 volume out) and for the device tile loop alone; 36 x 1,893.5 GFLOP of forward convs against the MFMA peak."""
    import fetal_net.model as fmodel
    from fetal_net.prediction import patch_wise_prediction
    patch = (64, 128, 128)
    if model is None:
        model = fmodel.unet_model_3d(input_shape=(1,) + patch)
    data = np.random.RandomState(0).randn(1, 160, 256, 256).astype(np.float32)
    out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5)      # captures the graphs
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5)
        times.append(time.perf_counter() - t0)
    st = model._tile_state
    sizes = sorted(st["per_b"], reverse=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    left = 36
    while left > 0:
        B = next(b for b in sizes if b <= left)
        pb = st["per_b"][B]
        pb["graph"].replay() if pb["graph"] is not None else pb["body"]()
        left -= B
    torch.cuda.synchronize()
    dev = time.perf_counter() - t0
    e2e = min(times)
    return {"workload": "configs[4]: 160x256x256 volume, patch 64x128x128, overlap_factor 0.5, 36 tiles, device tile groups %s, hipGraph" % sizes,
            "end_to_end_s_per_volume": e2e, "device_tile_loop_s": dev, "host_share": 1.0 - dev / e2e, "volumes_per_s": 1.0 / e2e,
            "fwd_tflops_device": 36 * 1893.5e9 / dev / 1e12, "mfma_frac_device": 36 * 1893.5e9 / dev / 1e12 / PEAK_BF16_TFLOPS,
            "finite": bool(np.isfinite(out).all())}


def secondary_line(cfg):
    """`bench.py --config cfg3|cfg4`: one JSON line for a secondary BASELINE configuration, same keys as the headline line"""
    torch.cuda.set_device(0)
    if cfg == "cfg3":
        r = cfg3_leg(steps=20, warmup=5)
        line = {"metric": "2D slices/sec (256x256x5, bf16) fwd+bwd", "value": r["slices_per_s"], "unit": "slices/s", "ms_per_step": r["ms_per_step"],
                "steps": r["steps"], "warmup": 5, "dtype": "bf16",
                "roofline": {"bound": "mfma", "achieved": r["conv_tflops_algorithmic"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": r["mfma_frac"],
                             "traffic": None, "note": "whole-step figure: all 3x3 conv FLOPs of a step / step time (not one kernel's launches)"}}
    else:
        r = cfg4_leg()
        line = {"metric": "sliding-window volumes/sec (160x256x256, patch 64x128x128, overlap 0.5, bf16)", "value": r["volumes_per_s"], "unit": "volumes/s",
                "ms_per_step": r["end_to_end_s_per_volume"] * 1e3, "steps": 3, "warmup": 1, "dtype": "bf16",
                "roofline": {"bound": "mfma", "achieved": r["fwd_tflops_device"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": r["mfma_frac_device"],
                             "traffic": None, "note": "device tile loop: 36 x 1,893.5 GFLOP of forward convs / loop time"}}
    line.update({"n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic", "config": {"workload": r["workload"]},
                 "detail": r})
    print(json.dumps(line))


TRAFFIC_PROFILE = os.path.join("profiles", "r06_pmc_traffic_per_step.json")
MFMA_PROFILE = os.path.join("profiles", "r06_pmc_mfma.json")


def kernel_source_hash():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, common.h, the Makefile with its per-file flags, the public header).  A profile under profiles/
    carries the hash of the tree it was collected on; the GPU box has no .git, so THIS is what bench.py can check at run time (the
    profile also records `git rev-parse HEAD` for the reader)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "fetal-mri-segmentation_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "fmri_hip.h")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key):
    """(HBM bytes per launch of the dominant kernel, stamp) from the committed rocprofv3 PMC passes (TRAFFIC_PROFILE: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same bench by tools/collect_profiles.sh; FETCH_SIZE doubled per the gfx950 note
    in MI355X_MICROARCH.md, HBM section).  The bytes are None when the file is missing OR was collected on other kernel sources than the
    ones in this tree (stamp["kernel_source_hash"] != kernel_source_hash()): a stale profile is not reported."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    stamp = {"file": TRAFFIC_PROFILE, "head": None, "kernel_source_hash": None, "current_kernel_source_hash": kernel_source_hash(), "fresh": False}
    if not os.path.exists(path):
        return None, stamp
    with open(path) as f:
        d = json.load(f)
    meta = d.get("_meta", {})
    stamp["head"], stamp["kernel_source_hash"] = meta.get("git_head"), meta.get("kernel_source_hash")
    stamp["fresh"] = stamp["kernel_source_hash"] == stamp["current_kernel_source_hash"]
    if not stamp["fresh"]:
        return None, stamp
    tot_b, calls = 0.0, 0
    for k, v in d.items():
        if k.startswith(kernel_key):
            tot_b += (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0
            calls += v["calls"]
    return ((tot_b / calls) if calls else None), stamp


def pmc_mfma(kernel_key):
    """MFMA utilisation of a kernel family from the committed counter pass (MFMA_PROFILE: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE,
    collected by tools/collect_profiles.sh in their own rocprofv3 --pmc run of this bench on one stream, summarised by tools/pmc_mfma.py):
    matrix-pipe busy cycles / (active cycles x 1024 SIMDs) - a clock-independent figure next to the FLOP-derived fraction of the 2.5 PF peak,
    which also carries clock / 2.4 GHz.  None when the file is missing or belongs to other kernel sources."""
    path = os.path.join(ROOT, MFMA_PROFILE)
    stamp = {"file": MFMA_PROFILE, "kernel_source_hash": None, "fresh": False}
    if not os.path.exists(path):
        return None, stamp
    with open(path) as f:
        d = json.load(f)
    stamp["kernel_source_hash"] = d.get("_meta", {}).get("kernel_source_hash")
    stamp["fresh"] = stamp["kernel_source_hash"] == kernel_source_hash()
    if not stamp["fresh"]:
        return None, stamp
    busy = act = 0.0
    for k, v in d.get("kernels", {}).items():
        if k.startswith(kernel_key):
            busy += v["mfma_busy_cycles_per_step"]
            act += v["active_cycles_per_step"]
    return ({"mfma_util": busy / (act * 1024.0), "mfma_busy_cycles_per_step": busy, "active_cycles_per_step": act,
             "counters": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), exclusive launches"} if act else None), stamp


def per_layer_table(eng, detail_ms, steps):
    """every 3x3x3 conv of the plan x (fwd, dgrad, wgrad): exclusive ms per launch, algorithmic TFLOP/s / 2.5 PF, algorithmic
    GB/s / 8 TB/s (bytes = |X| + |Y| + |W| in bf16, each tensor once: the pass reads two and writes one)."""
    p, N = eng.plan, eng.N
    first = p.enc[0][0]["name"]
    rows = []
    for c in p.convs_forward_order():
        D, H, W = p.level_dims(c["level"])
        vox = N * D * H * W
        fl = 2.0 * 27 * c["cin"] * c["cout"] * vox
        by = 2.0 * (vox * c["cin"] + vox * c["cout"] + 27 * c["cin"] * c["cout"])
        for ps in ("fwd", "dgrad", "wgrad"):
            if ps == "dgrad" and c["name"] == first:
                continue
            t = detail_ms.get((ps, c["cin"], c["cout"], D))
            if t is None:
                continue
            ms = t[0] / max(t[1], 1)
            rows.append({"layer": c["name"], "level": c["level"], "cin": c["cin"], "cout": c["cout"], "pass": ps, "ms": round(ms, 4),
                         "gflop": round(fl / 1e9, 1), "tflops": round(fl / (ms * 1e-3) / 1e12, 1),
                         "mfma_frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "algorithmic_mb": round(by / 1e6, 1), "gbs": round(by / (ms * 1e-3) / 1e9, 1),
                         "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                         "meets_40pct_hbm": bool(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS >= 0.40),
                         "parity_form": c["name"] in getattr(eng, "upcat", {})})
    tot_ms = sum(r["ms"] for r in rows)
    tot_fl = sum(r["gflop"] for r in rows)
    tot_by = sum(r["algorithmic_mb"] for r in rows)
    return {"rows": rows, "total": {"ms": round(tot_ms, 3), "gflop": round(tot_fl, 1), "mfma_frac": round(tot_fl / tot_ms / PEAK_BF16_TFLOPS, 4) if tot_ms else None,
                                    "algorithmic_mb": round(tot_by, 1), "hbm_frac": round(tot_by / tot_ms / PEAK_HBM_GBS, 4) if tot_ms else None},
            "note": "exclusive (one-stream) HIP-event time per launch, mean over %d steps; the network-level 40 %% HBM target is above the "
                    "MFMA ceiling (23.9 %%, SURVEY 0/8d)" % steps}


def launch_ranks(n, argv, port_tries=3):
    """Parent of a bare `bench.py --gpus N`: N children, one per GPU, each a rank of a 127.0.0.1 rendezvous.  The parent never touches
    the GPU (a process that has initialised HIP must not fork/exec workers on this pool).  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    if port_tries > 1:
        # the port is free now, not necessarily when rank 0 binds it (another job on the box may take it in between): a run that dies
        # on the rendezvous before printing anything is started again on a fresh port
        rc = 1
        for _ in range(port_tries):
            rc = launch_ranks(n, argv, port_tries=1)
            if rc != 98:
                return rc
        return rc
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is drained by a thread while the ranks run (a full pipe would block rank 0 inside a collective the others wait in)
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d - stopping the other ranks\n" % (r, code))
                for q in pending:                              # a dead rank leaves the others inside a collective
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    reader.join(timeout=10)
    line = None
    for ln in captured:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            sys.stderr.write(ln + "\n")
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    if line is not None and rc == 0:
        print(line)
    return rc


def selftest_cpu(a, rank, world):
    """Launcher / rendezvous / result-line plumbing on the CPU (`--selftest-cpu`, used by tests/test_bench_launcher.py): the ranks form a
    gloo group and all-reduce a small gradient-like buffer per "step"; nothing is measured and the line says so."""
    import torch.distributed as dist
    if world > 1:
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        except Exception as e:
            if rank == 0 and ("in use" in str(e).lower() or "EADDRINUSE" in str(e)):
                sys.exit(98)
            raise
        assert dist.get_world_size() == a.gpus
    g = torch.full((1 << 16,), float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        h = g.clone()
        if world > 1:
            dist.all_reduce(h)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert float(h[0]) == world * (world + 1) / 2.0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ranks = dist.get_world_size()
        dist.destroy_process_group()
    else:
        ranks = 1
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (CPU, gloo) - not a measurement", "value": a.batch * world * a.steps / dt,
                          "unit": "patches/s", "n_gpus": world, "collective_ranks": ranks, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "selftest", "config": {"workload": "selftest", "global_batch": a.batch * world,
                                                                           "parallelism": "dp%d" % world}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 50 steps = 0.8 s: the first ten run ~2 % slower than the steady state
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="untimed steps for at least this long before the --warmup steps (clock settling; not part of the timed region)")
    ap.add_argument("--lr", type=float, default=1e-4, help="learning rate of the timed steps (reference default 1e-4, fetal/config_utils.py:39); 0 keeps the "
                    "random-init weights as they are - a diagnostic: the power-limited shader clock depends on the operands' statistics")
    ap.add_argument("--data", choices=("learnable", "survey"), default="learnable",
                    help="learnable (default): a pool of batches of tools/learnable_task.py, walked round-robin; survey: rounds 1-3's ONE batch with labels "
                         "independent of the image (the net collapses to all-foreground and the backward kernels multiply ~0 gradients: a faster clock)")
    ap.add_argument("--pool", type=int, default=8, help="batches in the resident pool (learnable data)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--launch-timing-every", type=int, default=None,
                    help="HIP-event brackets on every n-th step of the timed region (1 = every step; they cost 2.3 %% of the step time when on every "
                         "step).  Default: 10, or 5 when --steps <= 20, so that `roofline` rests on at least four sampled steps")
    ap.add_argument("--config", choices=("cfg1", "cfg3", "cfg4"), default="cfg1",
                    help="cfg1 (default): the headline configs[1] training step.  cfg3 / cfg4: ONE line for BASELINE configs[3] (2-D training step) "
                         "or configs[4] (sliding-window inference of a 160x256x256 volume), N = 1")
    ap.add_argument("--val-dice-steps", type=int, default=150,
                    help="Adam steps of the val-Dice leg (after the timed region, N = 1 only; 0 = skip): see val_dice_leg")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short configs[3] / configs[4] / reference-API measurements that the default N = 1 line carries after the timed region")
    ap.add_argument("--no-exclusive-pass", action="store_true",
                    help="skip the second (single-stream) pass that measures exclusive kernel durations (used when profiling the timed region alone)")
    ap.add_argument("--serialize-streams", action="store_true",
                    help="run the conv weight gradients on the main stream (no concurrent kernels): per-kernel durations become exclusive")
    ap.add_argument("--per-layer", default=None, metavar="JSON",
                    help="also write the per-layer table (every 3x3x3 conv x fwd/dgrad/wgrad: exclusive ms, TFLOP/s / 2.5 PF, algorithmic "
                         "GB/s / 8 TB/s; SURVEY 8d) to this file")
    ap.add_argument("--selftest-cpu", action="store_true", help="exercise the N-rank launcher on the CPU (gloo); measures nothing")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.launch_timing_every is None:
        a.launch_timing_every = 5 if a.steps <= 20 else 10
    if a.config != "cfg1":
        if a.gpus != 1:
            ap.error("--config cfg3|cfg4 are single-GPU lines")
        return secondary_line(a.config)

    if a.gpus > 1 and "RANK" not in os.environ:
        # bare multi-GPU invocation: become the parent of N ranks BEFORE anything initialises the GPU
        # (no device-count probe here: on a ROCm build without amdsmi it is hipGetDeviceCount, which opens /dev/kfd in the parent; every
        # rank checks its own LOCAL_RANK against the visible devices and exits 2, which launch_ranks relays)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d\n" % (a.gpus, world))
        sys.exit(2)
    if a.selftest_cpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return selftest_cpu(a, rank, world)
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d: LOCAL_RANK %d but only %d device(s) visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dctx = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world)
        except Exception as e:                                  # rendezvous port taken between the parent's probe and rank 0's bind
            if rank == 0 and ("in use" in str(e).lower() or "EADDRINUSE" in str(e)):
                sys.stderr.write("bench.py: rendezvous port busy: %s\n" % e)
                sys.exit(98)
            raise
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        from fmri_hip.dist import DataParallel
        dctx = DataParallel(world, rank)

    from fmri_hip import ops
    from fmri_hip.engine import UNetEngine, UNetPlan

    spatial = (64, 128, 128)
    plan = UNetPlan(1, spatial, depth=4, n_base_filters=32)
    eng = UNetEngine(plan, a.batch, dtype=torch.bfloat16, dist_ctx=dctx, seed=42)
    if dctx is not None:
        dctx.broadcast_params(eng)
    if a.serialize_streams:
        eng._wg_stream = None
    # Input data (round 4): a pool of batches of the LEARNABLE task (tools/learnable_task.py), resident in HBM, walked round-robin, trained on
    # at the reference's learning rate - a network that is really learning.  Rounds 1-3 trained the timed steps on ONE batch whose labels were
    # independent of the image (SURVEY 8d's recipe, 30 % foreground): with the reference's Dice loss that network falls into the
    # all-foreground solution within ~25 steps (train Dice 0.4615 = 2 fg / (1 + fg)), the sigmoid saturates, the gradients that the
    # input-gradient and weight-gradient kernels multiply are ~0 - and MI355X's power-limited shader clock rises: the SAME kernels ran 13-15 %
    # faster (313 against 272 patches/s on one box: profiles/r04_data_dependence.json).  That state says nothing about training; `value` is
    # now measured on live data.  --data survey restores the old recipe (the `continuity` object of the line reports it as well).
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT

    def make_pool(kind, n):
        pool = []
        for k in range(n):
            if kind == "survey":
                x, y = synthetic_batch((a.batch, 1) + spatial, seed_x=1234 + rank + 97 * k, seed_y=1235 + rank + 97 * k)
                xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
            else:
                xt, yt = LT.device_batch(1_000 * rank + k * a.batch, a.batch, spatial)
            pool.append((xt.to(torch.bfloat16).reshape(a.batch, *spatial, 1).contiguous(), yt.reshape(-1).contiguous()))
        return pool

    pool = make_pool(a.data, 1 if a.data == "survey" else a.pool)
    nxt = [0]

    def step():
        xd, yd = pool[nxt[0] % len(pool)]
        nxt[0] += 1
        return eng.train_step(xd, yd, lr)

    lr = a.lr

    timer = LaunchTimer()
    if not a.no_launch_timing:
        # detail key = (pass, Cin of the conv, Cout of the conv, D of its output): unique per conv of the plan
        def lab_fwd(src0, src1, w, bias, y_, *aa, **kk):
            c0, c1 = src0.shape[-1], (0 if src1 is None else src1.shape[-1])
            key = ("fwd", c0 + c1, y_.shape[-1], y_.shape[1])
            if c0 == 1 and c1 == 0:
                return "conv_first_fwd", key
            return ("conv_fwd_mfma" if (c0 % 32 == 0 and c1 % 32 == 0 and y_.shape[-1] % 32 == 0) else "conv_fwd_generic"), key

        def lab_dgrad(dy, wd, dx, *aa, **kk):
            key = ("dgrad", dx.shape[-1], dy.shape[-1], dy.shape[1])
            return ("conv_fwd_mfma" if (dy.shape[-1] % 32 == 0 and dx.shape[-1] % 32 == 0) else "conv_fwd_generic"), key

        def lab_wgrad(src0, src1, dy, dw, db, *aa, **kk):
            c0, c1 = src0.shape[-1], (0 if src1 is None else src1.shape[-1])
            key = ("wgrad", c0 + c1, dy.shape[-1], dy.shape[1])
            if c0 == 1 and c1 == 0:
                return "conv_first_wgrad", key
            return ("conv_wgrad_mfma" if (c0 % 32 == 0 and c1 % 32 == 0 and dy.shape[-1] % 64 == 0) else "conv_wgrad_generic"), key

        def lab_up_fwd(x_low, skip, w_up, w_sk, bias, y_, *aa, **kk):
            return "conv_fwd_mfma", ("fwd", x_low.shape[-1] + (0 if skip is None else skip.shape[-1]), y_.shape[-1], y_.shape[1])

        def lab_up_dgrad(dy, *aa, **kk):
            # the parity-form input gradient covers BOTH halves of the concat (low-res tensor + skip tensor) and the up-sampling gradient
            dlow, dskip = aa[4], aa[5]
            return "conv_fwd_mfma", ("dgrad", dlow.shape[-1] + (0 if dskip is None else dskip.shape[-1]), dy.shape[-1], dy.shape[1])

        def lab_up_wgrad(x_low, skip, dy, *aa, **kk):
            return "conv_wgrad_mfma", ("wgrad", x_low.shape[-1] + (0 if skip is None else skip.shape[-1]), dy.shape[-1], dy.shape[1])

        def lab_tail(src0, w, bias, y_, *aa, **kk):      # conv block whose epilogue also pools / computes the final 1x1x1 conv
            return "conv_fwd_mfma", ("fwd", src0.shape[-1], y_.shape[-1], y_.shape[1])

        timer.wrap(ops, "conv3d_fwd", lab_fwd)
        timer.wrap(ops, "conv3d_fwd_tail", lab_tail)
        timer.wrap(ops, "conv3d_dgrad", lab_dgrad)
        timer.wrap(ops, "conv3d_wgrad", lab_wgrad)
        timer.wrap(ops, "conv3d_upcat_fwd", lab_up_fwd, launches=2)
        timer.wrap(ops, "conv3d_upcat_dgrad", lab_up_dgrad, launches=2)
        # (FMRI_PACK_BATCHED=0 only:) the optimizer step repacks the deep layers' weight images on a side stream, under the next step's first
        # convolutions: their event brackets measure a concurrent span, not a cost on the step's critical path
        def lab_pack(*aa, **kk):
            return "pack_weights" if torch.cuda.current_stream() == torch.cuda.default_stream() else "pack_weights_side_stream_overlapped"

        timer.wrap(ops, "conv3d_pack_up_weights", lab_pack)
        timer.wrap(ops, "conv3d_upcat_wgrad", lab_up_wgrad, launches=2)
        timer.wrap(ops, "pack_weights", lab_pack)
        timer.wrap(ops, "pack_weights_batched", lambda *aa, **kk: "pack_weights_batched")       # round 6: one launch for all images, main stream
        for nm in ("maxpool_fwd", "maxpool_bwd", "upsample_bwd", "conv1x1_fwd", "conv1x1_bwd", "sigmoid_dice_fwd",
                   "sigmoid_dice_bwd", "adam_step"):
            timer.wrap(ops, nm, (lambda n_: (lambda *aa, **kk: n_))(nm))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # Settled clock (SURVEY 8d ">= 2 s warm"; VERDICT r2: the driver's 5 warm-up steps are 74 ms): untimed steps for at least
    # --prewarm-seconds of wall time BEFORE the W requested warm-up steps - the same number of steps on every rank (rank 0 decides)
    prewarm_steps = 0
    if a.prewarm_seconds > 0:
        t_pw = time.perf_counter()
        while True:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            prewarm_steps += 10
            go_on = torch.tensor([1.0 if time.perf_counter() - t_pw < a.prewarm_seconds else 0.0], device="cuda")
            if world > 1:
                import torch.distributed as dist
                dist.broadcast(go_on, 0)
            if float(go_on.item()) == 0.0:
                break
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # HIP-event brackets inside the timed region are instrumentation, not work, and they are not free: an event pair around every op of
    # every step cost 0.3 ms per step = 2.3 % (interleaved same-box runs: 298-304 vs 306-311 patches/s without any).  The cost sits on the
    # MAIN stream - each event is one more packet in the chain of dependent input-gradient launches that the step's length hangs on
    # (events around the weight-gradient stream's ops alone: 310-311, i.e. free; around the main stream's conv ops alone: 287-293, the
    # main stream then falls behind the other).  So the launch durations are SAMPLED: every --launch-timing-every-th step of the timed
    # region (default 10) carries the brackets, around every op as before; `roofline` is computed from those steps' launches
    # (`roofline.sampled_steps`), `value` from the wall clock over all K steps.
    every = max(1, a.launch_timing_every)
    sampled = 0
    # shader clock of the timed region: two {s_memtime, s_memrealtime} stamps per XCD on the main stream, one launch
    # of 64 one-wave workgroups each - outside the K steps' kernels, inside the synchronised bracket (2 x ~5 us of 0.7 s)
    stamp0, stamp1 = torch.zeros(16, dtype=torch.int64, device="cuda"), torch.zeros(16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.clock_stamp(stamp0)
    for i in range(a.steps):
        timer.on = (not a.no_launch_timing) and (i % every == 0)
        sampled += 1 if timer.on else 0
        sums = step()
    ops.clock_stamp(stamp1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tot_timed = timer.totals_ms()
    detail = timer.detail_ms()          # exclusive only when everything ran on one stream (replaced by the second pass below otherwise)
    tot_excl = None
    if not a.no_launch_timing and not a.no_exclusive_pass and eng._wg_stream is not None:
        # The engine runs the weight-gradient kernels on a second stream, concurrently with the input-gradient chain: inside the timed
        # region a kernel's event bracket (and its rocprofv3 duration) includes the time it shares the CUs with the other stream's kernel.
        # A second pass of the same K steps with that stream switched off gives the EXCLUSIVE durations (every rank runs it: the steps
        # contain the gradient all-reduce).
        keep, eng._wg_stream = eng._wg_stream, None
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        timer.rec, timer.detail = {}, {}
        timer.on = True
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        timer.on = False
        tot_excl = timer.totals_ms()
        detail = timer.detail_ms()
        eng._wg_stream = keep

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt
    m = eng.metrics_from_sums(sums.cpu().numpy())
    out = {
        "metric": "3D patches/sec (64x128x128, bf16) fwd+bwd", "value": value, "unit": "patches/s", "n_gpus": world,
        "collective_ranks": (dctx.world if dctx is not None else 1),
        "steps": a.steps, "warmup": a.warmup, "prewarm_steps_untimed": prewarm_steps, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16",
        "data": ("synthetic (learnable task: image and label from one latent field, %d resident batches walked round-robin, lr %g; random-init weights)" % (len(pool), lr))
        if a.data == "learnable" else "synthetic (SURVEY 8d recipe: one batch, labels independent of the image; lr %g)" % lr,
        "config": {"workload": "configs[1]: depth-4 3D U-Net, 32 base filters, bf16, batch %dx1x64x128x128 per GPU, "
                               "full step = fwd + Dice + bwd + Keras-Adam" % a.batch,
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world},
        "train_dice_last_step": m["dice_coefficient"],
        "hbm_roofline_frac_conv_algorithmic": (value / world) * ALGO_BYTES_PER_PATCH / 1e9 / PEAK_HBM_GBS,
    }
    ghz, per_xcd = ops.clock_ghz(stamp0, stamp1)
    out["clock_ghz"] = None if ghz is None else round(ghz, 3)
    out["clock_ghz_note"] = ("average shader clock over the timed region of rank 0: d(s_memtime) / d(s_memrealtime) x 0.1 GHz between two stamps on the "
                             "main stream, median over the XCDs (per XCD: %s); the chip is power-limited under this workload, 2.4 GHz is the "
                             "clock behind the 2.5 PF peak" % " ".join("%.3f" % v for v in per_xcd))
    if not a.no_launch_timing:
        fl = conv_flops(eng)

        def roofline_of(tot, note, nsteps):
            per_step = {k: (v[0] / nsteps, v[1] // nsteps) for k, v in tot.items()}
            dom = max(("conv_fwd_mfma", "conv_wgrad_mfma"), key=lambda k: per_step.get(k, (0, 0))[0])
            if dom not in per_step:
                return per_step, None, None
            flops = (fl["fwd_mfma"] + fl["dgrad_mfma"]) if dom == "conv_fwd_mfma" else fl["wgrad_mfma"]
            t_ms, launches = per_step[dom]
            ach = flops / (t_ms * 1e-3) / 1e12
            traffic, stamp = pmc_traffic("k_conv_fwd_" if dom == "conv_fwd_mfma" else "k_conv_wgrad")   # every instantiation of the family
            util, ustamp = pmc_mfma("k_conv_fwd_" if dom == "conv_fwd_mfma" else "k_conv_wgrad")
            r = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                 "note": note,
                 "traffic": traffic, "traffic_profile": stamp, "mfma_util_pmc": util, "mfma_util_profile": ustamp,
                 "traffic_source": "COMMITTED PROFILE, not this run: %s (a separate rocprofv3 --pmc pass of this bench, tools/collect_profiles.sh); "
                                   "reported only while that profile's kernel_source_hash equals this tree's, else null" % TRAFFIC_PROFILE,
                 "mfma_util_source": "COMMITTED PROFILE, not this run: %s" % MFMA_PROFILE,
                 "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, separate passes, FETCH_SIZE x2 corrected); null when the committed "
                                 "profile was collected on other kernel sources",
                 "algorithmic_flop_per_launch": flops / max(launches, 1), "launches_per_step": launches,
                 "avg_launch_ms": t_ms / max(launches, 1), "sampled_steps": nsteps}
            if dom == "conv_fwd_mfma":
                r["executed_tflops"] = fl["fwd_dgrad_executed"] / (t_ms * 1e-3) / 1e12
            other = "conv_wgrad_mfma" if dom == "conv_fwd_mfma" else "conv_fwd_mfma"
            ro = None
            if other in per_step:
                fo = (fl["fwd_mfma"] + fl["dgrad_mfma"]) if other == "conv_fwd_mfma" else fl["wgrad_mfma"]
                ro = {"kernel": other, "achieved": fo / (per_step[other][0] * 1e-3) / 1e12, "unit": "TFLOP/s"}
            return per_step, r, ro

        base_note = ("achieved = the reference's algorithmic FLOPs (2*27*Cin*Cout*voxels per conv, SURVEY 8d) / kernel time; the decoder 'a' "
                     "layers run in parity form (8 instead of 27 taps on the up-sampled channels), executed_tflops counts the MACs really issued")
        concurrent = tot_excl is not None
        per_step, r, ro = roofline_of(tot_timed, base_note + ("; TIMED REGION WITH TWO STREAMS: the weight-gradient kernels run concurrently with "
                                                              "this kernel, so the HIP-event bracket of a launch is not exclusive time (see "
                                                              "roofline_exclusive)" if concurrent else ""), max(sampled, 1))
        out["kernel_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0])}
        if concurrent:
            out["kernel_ms_per_step_note"] = "event brackets of concurrently running kernels overlap: the sum exceeds ms_per_step"
        if r is not None:
            out["roofline"] = r
        if ro is not None:
            out["roofline_other"] = ro
        if concurrent:
            per_x, rx, rox = roofline_of(tot_excl, base_note + "; second pass of the same K steps with the weight-gradient stream switched off "
                                                               "(= bench.py --serialize-streams): exclusive kernel durations", a.steps)
            if rx is not None:
                out["roofline_exclusive"] = rx
                out["roofline_exclusive"]["kernel_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(per_x.items(), key=lambda kv: -kv[1][0])}
            if rox is not None:
                out["roofline_exclusive"]["other"] = rox
        # step-level figure that does not depend on how kernels overlap: all conv FLOPs of a step / the step time
        out["step_mfma_frac"] = (fl["fwd_mfma"] + fl["dgrad_mfma"] + fl["wgrad_mfma"]) / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
    if a.per_layer and not a.no_launch_timing:
        tab = per_layer_table(eng, detail, a.steps)
        tab["exclusive"] = bool(tot_excl is not None or a.serialize_streams or eng._wg_stream is None)
        tab["kernel_source_hash"] = kernel_source_hash()
        with open(a.per_layer, "w") as f:
            json.dump(tab, f, indent=1)
        out["per_layer_total"] = tab["total"]
    if world == 1 and (a.val_dice_steps > 0 or not a.no_secondary):
        # after the timed region, never inside it: the bench engine's buffers go first (each leg builds its own model)
        if a.data == "learnable" and not a.no_secondary:
            # continuity with rounds 1-3: the same engine, re-initialised, on the old recipe (one batch, independent labels), >= 2 s of
            # untimed steps (by then it has collapsed and the clock has settled), then 50 timed steps (round 4: 1.5 s + 20 steps scattered
            # 306-337 patches/s from box to box)
            eng.init_glorot(42)
            eng.M.zero_()
            eng.V.zero_()
            eng.t = 0
            xo, yo = make_pool("survey", 1)[0]
            t_c = time.perf_counter()
            while time.perf_counter() - t_c < 2.0:
                for _ in range(10):
                    eng.train_step(xo, yo, 1e-4)
                torch.cuda.synchronize()
            n_c = 50
            t_c = time.perf_counter()
            ops.clock_stamp(stamp0)
            for _ in range(n_c):
                so = eng.train_step(xo, yo, 1e-4)
            ops.clock_stamp(stamp1)
            torch.cuda.synchronize()
            dt_c = time.perf_counter() - t_c
            ghz_c, _ = ops.clock_ghz(stamp0, stamp1)
            out["continuity"] = {"recipe": "rounds 1-3: ONE batch, labels independent of the image (SURVEY 8d), trained on at lr 1e-4 - the net sits in the "
                                           "all-foreground solution of the Dice loss, the backward kernels multiply ~0 gradients, the power-limited clock rises",
                                 "value": n_c * a.batch / dt_c, "unit": "patches/s", "steps": n_c, "clock_ghz": None if ghz_c is None else round(ghz_c, 3),
                                 "train_dice_last_step": eng.metrics_from_sums(so.cpu().numpy())["dice_coefficient"]}
            del xo, yo
        del eng, pool
        torch.cuda.empty_cache()
        model = None
        if a.val_dice_steps > 0:
            out["val_dice"], model = val_dice_leg(a.batch, a.val_dice_steps)
        if not a.no_secondary:
            import fetal_net.metrics as FM
            import fetal_net.model as fmodel
            if model is None:
                model = fmodel.unet_model_3d(input_shape=(1, 64, 128, 128), depth=4, n_base_filters=32, initial_learning_rate=1e-4,
                                             loss_function=FM.dice_coefficient_loss)
            out["reference_api"] = reference_api_leg(model, a.batch)
            out["secondary"] = {"cfg4": cfg4_leg(model), "cfg3": cfg3_leg()}
            del model
            torch.cuda.empty_cache()
    ref = None
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"], ref = cpu_baseline()
    if world == 1 and not a.no_secondary:
        out["parity_mode"] = parity_mode_leg(a.batch, ref)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
