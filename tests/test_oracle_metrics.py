"""Oracle metrics vs the reference's own outputs (tests/golden/metrics_golden.json, incl. test/test_metrics.py KATs)."""
import json
import os

import numpy as np
import pytest

from oracle import metrics_oracle as M


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "metrics_golden.json")) as f:
        return json.load(f)


def _case_inputs(c):
    rs = np.random.RandomState(c["seed"])
    y = (rs.rand(*c["shape"]) > c["thr"]).astype(np.float32)
    p = rs.rand(*c["shape"]).astype(np.float32)
    return y.astype(np.float64), p.astype(np.float64)


def test_seeded_cases(gold):
    for c in gold["cases"]:
        y, p = _case_inputs(c)
        for key, fn in [("dice", M.dice_coefficient), ("dice_loss", M.dice_coefficient_loss), ("vod", M.vod_coefficient),
                        ("vod_loss", M.vod_coefficient_loss), ("weighted_dice", M.weighted_dice_coefficient),
                        ("weighted_dice_loss", M.weighted_dice_coefficient_loss), ("double_dice_loss", M.double_dice_loss),
                        ("dice_and_xent", M.dice_and_xent), ("weighted_cross_entropy", M.weighted_cross_entropy_loss),
                        ("focal_loss", M.focal_loss)]:
            # vod: the reference casts the binarised masks to float32 (tf.as_dtype(float)) -> fp32 sums
            rel = 1e-6 if key == "vod" else 1e-12
            assert fn(y, p) == pytest.approx(c[key], rel=rel, abs=1e-12), (c["seed"], key)


def test_reference_kats(gold):
    # reference test/test_metrics.py:10-38
    k = gold["weighted_dice_kat"]
    data = np.zeros((5 ** 3) * 3).reshape(3, 5, 5, 5)
    data[0, 0:1] = 1
    data[1, 0:2] = 1
    data[2, 1:4] = 1
    mx = M.weighted_dice_coefficient(data, data)
    assert mx == pytest.approx(k["max_dice"], abs=1e-12)
    for i in range(3):
        t = data.copy()
        t[i] = 0
        d = M.weighted_dice_coefficient(data, t)
        assert d == pytest.approx(k["drop_channel"][i], abs=1e-12)
        assert abs(d - 2 * mx / 3) < 1e-5
    assert abs(M.weighted_dice_coefficient(data, np.zeros_like(data))) < 1e-5
    d2 = np.zeros_like(data)
    d2[1, 0:2] = 1
    d2[2, 1:4] = 1
    assert M.weighted_dice_coefficient(d2, d2) == 1
