"""Val Dice - the second leg of BASELINE.json's metric (SURVEY 8d "val Dice" (i) soft Dice = -val_loss, reference fetal_net/metrics.py:11-15;
(ii) hard Dice of p > 0.5, reference fetal/evaluate.py:16-17, over a patch_wise_prediction-reconstructed volume).

Does bf16 training on the MFMA kernels CONVERGE to the Dice the fp32 path and the CPU oracle reach?  The same learnable synthetic task
(tools/learnable_task.py), the same initial weights (Keras glorot, seed 42), the same batches, 100 Adam steps through the reference's
own `train_model()` (callbacks, checkpoints, CSV log and all), depth 4 / 32 base filters at 32x64x128 so that every conv of the bf16 run
is on the benchmarked MFMA kernels:

    bf16 engine  vs  fp32 engine (VALU kernels)  vs  oracle (torch-CPU fp32, Keras-Adam; tests/golden/val_dice_oracle.json, written by
    tests/golden/make_val_dice_fixture.py - FMRI_LIVE_ORACLE=1 runs that function on the box instead of reading the file)

Bars: set from measured runs on MI355X (FMRI_MEASURE=1 records instead of asserting: profiles/r04_val_dice_runs.jsonl holds them).
Training is a chaotic map - two fp32 runs that differ in summation order drift apart step by step - so the bars are on the Dice
the runs END at (and per epoch), not on weights.
"""
import glob
import itertools
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from gpu_util import bar  # noqa: E402


def _oracle_numbers():
    import make_val_dice_fixture as MK
    if os.environ.get("FMRI_LIVE_ORACLE", "0") == "1":
        return MK.run_oracle()
    with open(os.path.join(ROOT, "tests", "golden", "val_dice_oracle.json")) as f:
        ref = json.load(f)
    cfg = ref["config"]
    assert (tuple(cfg["spatial"]), cfg["batch"], cfg["epochs"], cfg["steps_per_epoch"], cfg["validation_steps"], cfg["lr"]) == \
        (MK.SPATIAL, MK.BATCH, MK.EPOCHS, MK.STEPS, MK.VAL, MK.LR), "fixture was generated for another schedule: re-run make_val_dice_fixture.py"
    return ref


def _train_through_the_reference_api(dtype, tmp_path):
    """-> dict(per-epoch held-out soft Dice, per-step training loss, final held-out soft / hard Dice, volume hard / soft Dice)"""
    import fetal_net.metrics as FM
    import fetal_net.model as fmodel
    import learnable_task as LT
    import make_val_dice_fixture as MK
    from fetal_net.engine_model import Callback
    from fetal_net.prediction import patch_wise_prediction
    from fetal_net.training import train_model
    from oracle import unet_oracle as O
    out_dir = tmp_path / dtype
    out_dir.mkdir()
    model = fmodel.unet_model_3d(input_shape=(1,) + MK.SPATIAL, depth=4, n_base_filters=32, initial_learning_rate=MK.LR,
                                 loss_function=FM.dice_coefficient_loss, compute_dtype=dtype)
    model.set_weights_dict(O.Spec((1,) + MK.SPATIAL, depth=4, n_base_filters=32).init_weights(MK.SEED_W))
    held = [LT.host_batch(LT.HELD_OUT + k * MK.BATCH, MK.BATCH, MK.SPATIAL) for k in range(MK.VAL)]

    class StepLosses(Callback):                                    # reads every batch log (forces the deferred reads: fine for a test)
        def __init__(self):
            self.loss = []

        def on_batch_end(self, batch, logs=None):
            self.loss.append(float(logs["loss"]))

    # train_model builds its own callback list; the per-step losses come from a second pass-through callback on the model's History
    steps = StepLosses()
    orig_fit = model.fit_generator

    def fit_with_probe(**kw):
        kw["callbacks"] = list(kw.get("callbacks") or []) + [steps]
        return orig_fit(**kw)

    model.fit_generator = fit_with_probe
    hist = train_model(model, str(out_dir / "fetal_net_model"), LT.host_generator(0, MK.BATCH, MK.SPATIAL), itertools.cycle(held),
                       steps_per_epoch=MK.STEPS, validation_steps=MK.VAL, initial_learning_rate=MK.LR, n_epochs=MK.EPOCHS,
                       output_folder=str(out_dir)).history
    assert len(hist["val_loss"]) == MK.EPOCHS and len(steps.loss) == MK.EPOCHS * MK.STEPS
    assert glob.glob(str(out_dir / "fetal_net_model") + "*.h5"), "ModelCheckpoint wrote no best-val_loss file"
    P = [model.predict(x) for x, _ in held]
    res = dict(val_soft_dice_per_epoch=[-float(v) for v in hist["val_loss"]], train_loss=steps.loss,
               held_out_soft_dice=float(np.mean([LT.soft_dice(y, p) for (_, y), p in zip(held, P)])),
               held_out_hard_dice=float(np.mean([LT.hard_dice(y, p > 0.5) for (_, y), p in zip(held, P)])))
    # the Keras log's val_loss of the last epoch IS the held-out soft Dice (same batches, evaluation mode)
    assert abs(res["val_soft_dice_per_epoch"][-1] - res["held_out_soft_dice"]) <= 1e-5
    vx, vy = LT.host_patch(LT.HELD_OUT + MK.VOLUME_SEED_OFFSET, MK.VOLUME)
    rec = patch_wise_prediction(model, vx[None].astype(np.float64), MK.SPATIAL, overlap_factor=MK.OVERLAP)
    assert rec.shape == MK.VOLUME + (1,) and rec.dtype == np.float64
    res["volume_hard_dice"] = LT.hard_dice(vy, rec[..., 0] > 0.5)
    res["volume_soft_dice"] = LT.soft_dice(vy, rec[..., 0])
    return res


def test_bf16_training_reaches_the_dice_of_fp32_and_of_the_cpu_oracle(tmp_path):
    ref = _oracle_numbers()
    runs = {"oracle": ref}
    for dtype in ("fp32", "bf16"):
        runs[dtype] = _train_through_the_reference_api(dtype, tmp_path)
        torch.cuda.synchronize()
    if os.environ.get("FMRI_MEASURE", "0") == "1":
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "val_dice_runs.jsonl")
        with open(path, "a") as f:
            f.write(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "config"} for k, v in runs.items()}) + "\n")
    # 1. the task is learned: far above the all-foreground / base-rate solutions the unlearnable bench batch allows
    for who in ("oracle", "fp32", "bf16"):
        assert runs[who]["held_out_soft_dice"] >= 0.92, (who, runs[who]["held_out_soft_dice"])          # measured 0.937 - 0.941
        assert runs[who]["volume_hard_dice"] >= 0.93, (who, runs[who]["volume_hard_dice"])              # measured 0.948 - 0.951
    # 2. parity of the END POINT: soft Dice (= -val_loss), hard Dice on held-out batches, hard Dice of the reconstructed volume
    for a, b in (("bf16", "fp32"), ("bf16", "oracle"), ("fp32", "oracle")):
        for key, limit in (("held_out_soft_dice", BARS[(a, b)][0]), ("held_out_hard_dice", BARS[(a, b)][1]), ("volume_hard_dice", BARS[(a, b)][2])):
            bar("val dice %s vs %s: |d %s|" % (a, b, key), abs(runs[a][key] - runs[b][key]), limit)
    # 3. trajectory: per-epoch held-out soft Dice and the per-step training loss stay together (reported; loose bars - chaotic map)
    for a, b in (("bf16", "fp32"), ("bf16", "oracle"), ("fp32", "oracle")):
        d_ep = max(abs(x - y) for x, y in zip(runs[a]["val_soft_dice_per_epoch"], runs[b]["val_soft_dice_per_epoch"]))
        d_st = max(abs(x - y) for x, y in zip(runs[a]["train_loss"], runs[b]["train_loss"]))
        bar("val dice %s vs %s: max per-epoch |d soft dice|" % (a, b), d_ep, BARS[(a, b)][3])
        bar("val dice %s vs %s: max per-step |d train loss|" % (a, b), d_st, BARS[(a, b)][4])


# (held-out soft, held-out hard, volume hard, per-epoch soft, per-step train loss): <= 2x the largest value over six measured runs on MI355X
# (FMRI_MEASURE=1; the last four are profiles/r04_val_dice_runs.jsonl).  Measured maxima: end points 3.5e-3 / 3.5e-3 / 2.5e-3 for every
# pair (two runs of ONE engine differ by as much: fp32 atomics order -> chaotic map); per epoch 3.0e-2 with bf16 (epoch 1, where the
# curve is steepest: bf16 0.82 vs 0.79), 4.5e-3 fp32 vs oracle; per step 6.4e-2 with bf16, 1.05e-2 fp32 vs oracle.
BARS = {("bf16", "fp32"): (7e-3, 7e-3, 5e-3, 6e-2, 1.3e-1),
        ("bf16", "oracle"): (7e-3, 7e-3, 5e-3, 6e-2, 1.3e-1),
        ("fp32", "oracle"): (5e-3, 5e-3, 7e-3, 9e-3, 2.1e-2)}
# (round 5: the volume's hard Dice of fp32 vs the oracle came out 4.1e-3 apart in one of ~15 further runs of unchanged code - the bar was 3.5e-3 = 2x
# the maximum of the first six; a thresholded volume moves in steps and the fp32 engine's atomics reorder every run.  Only that thresholded
# bar is widened (7e-3 = 2x the largest end-point difference seen between ANY two runs); the two held-out bars of the fp32 pair stay at 5e-3)
