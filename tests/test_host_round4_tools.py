"""CPU checks of the round-4 measurement tooling: the learnable synthetic task (tools/learnable_task.py: determinism, foreground fraction,
held-out split, the reference's two Dice definitions), the device-code hash of the built library (tools/lib_code_hash.py) and the profile
summary generator (tools/summarize_profiles.py)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import learnable_task as LT  # noqa: E402


def test_a_patch_is_a_pure_function_of_its_seed_and_has_the_stated_foreground():
    sp = (16, 32, 32)
    x0, y0 = LT.host_patch(7, sp)
    x1, y1 = LT.host_patch(7, sp)
    assert np.array_equal(x0, x1) and np.array_equal(y0, y1)
    x2, y2 = LT.host_patch(8, sp)
    assert not np.array_equal(y0, y2)
    assert x0.dtype == np.float32 and y0.dtype == np.uint8 and set(np.unique(y0)) <= {0, 1}
    assert abs(float(y0.mean()) - LT.FG) < 2e-3                       # the label is the (1 - fg) quantile of the latent field
    assert abs(float(x0.mean())) < 1e-5 and abs(float(x0.std()) - 1.0) < 1e-4        # z-scored like the reference's volumes
    # the image carries the label: foreground voxels are brighter by ~contrast noise sigmas
    gap = float(x0[y0 == 1].mean() - x0[y0 == 0].mean()) * float(np.sqrt(1 + LT.CONTRAST ** 2 * LT.FG * (1 - LT.FG)))
    assert abs(gap - LT.CONTRAST) < 0.2


def test_batches_follow_the_reference_generator_contract_and_the_split_is_disjoint():
    sp = (8, 16, 16)
    x, y = LT.host_batch(0, 3, sp)
    assert x.shape == (3, 1) + sp and x.dtype == np.float64 and y.shape == (3, 1) + sp and y.dtype == np.uint8      # generator.py:397-401
    g = LT.host_generator(0, 2, sp, steps=2)
    b0, b1 = next(g), next(g)
    with pytest.raises(StopIteration):
        next(g)
    assert np.array_equal(b0[0][1], LT.host_batch(1, 1, sp)[0][0]) and np.array_equal(b1[0][0], LT.host_batch(2, 1, sp)[0][0])
    held = LT.host_batch(LT.HELD_OUT, 2, sp)
    assert not any(np.array_equal(held[1][i], LT.host_batch(s, 1, sp)[1][0]) for i in range(2) for s in range(8))


def test_dice_definitions_are_the_references():
    from oracle import metrics_oracle as MO
    rs = np.random.RandomState(0)
    y = (rs.rand(2, 1, 8, 8, 8) > 0.7).astype(np.float64)
    p = rs.rand(2, 1, 8, 8, 8)
    assert LT.soft_dice(y, p) == pytest.approx(float(MO.dice_coefficient(y, p)), abs=1e-12)           # metrics.py:11-15
    t, q = y[0, 0], (p[0, 0] > 0.5)
    assert LT.hard_dice(t, q) == pytest.approx(2.0 * (t * q).sum() / (t.sum() + q.sum()), abs=1e-15)   # fetal/evaluate.py:16-17
    assert LT.hard_dice(t, t) == 1.0


def test_the_oracle_fixture_of_the_val_dice_test_belongs_to_the_current_task():
    """tests/golden/val_dice_oracle.json was generated with the task's constants and the schedule the GPU test uses"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_val_dice_fixture as MK
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "val_dice_oracle.json")))
    cfg = ref["config"]
    assert (tuple(cfg["spatial"]), cfg["batch"], cfg["epochs"], cfg["steps_per_epoch"], cfg["validation_steps"]) == (MK.SPATIAL, MK.BATCH, MK.EPOCHS, MK.STEPS, MK.VAL)
    assert len(ref["train_loss"]) == MK.EPOCHS * MK.STEPS and len(ref["val_soft_dice_per_epoch"]) == MK.EPOCHS
    assert ref["held_out_soft_dice"] > 0.92 and ref["volume_hard_dice"] > 0.93                        # the oracle learns the task
    # the first training loss is what the random-init network gives on seed 0: Dice of p ~ 0.5 against 12 % foreground
    assert -ref["train_loss"][0] == pytest.approx(2 * 0.5 * LT.FG / (LT.FG + 0.5), abs=0.02)


def test_device_code_hash_is_recorded_and_stable():
    import lib_code_hash
    so = os.path.join(ROOT, "fetal-mri-segmentation_amd", "lib", "libfmri_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built here")
    h, n = lib_code_hash.code_hash(so)
    assert n == 11 and len(h) == 64                                    # one gfx950 code object per .hip file
    assert h == lib_code_hash.code_hash(so)[0]
    assert open(so + ".sha256").read().strip() == h


def test_profile_summary_generator_reads_every_file_kind():
    import summarize_profiles as SP
    text = SP.summary("r04")
    for needle in ("Per layer", "conv3d_14", "one stream", "HBM traffic per step", "Matrix-pipe utilisation", "default `python bench.py`", "`continuity`",
                   "`val_dice`", "`reference_api`", "`secondary`", "`cpu_baseline`"):
        assert needle in text, needle
    assert SP.demote(text).startswith("#### Per layer") or SP.demote(text).lstrip().startswith("#### Per layer")
