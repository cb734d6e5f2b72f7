"""Oracle tiler vs outputs of the reference's prediction.py / patches.py (tests/golden/tiler_golden.*)."""
import json
import os

import numpy as np
import pytest

from oracle import tiler_oracle as T


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "tiler_golden.json")) as f:
        meta = json.load(f)
    arrs = np.load(os.path.join(golden_dir, "tiler_golden.npz"))
    return meta, arrs


class FakeModel3D:
    def __init__(self, patch, n_out=1):
        self.patch = tuple(patch)
        self.output_shape = (None, n_out) + self.patch
        self.n_out = n_out
        g = np.meshgrid(*[np.arange(s) for s in self.patch], indexing="ij")
        self.ramp = (g[0] * 1.0 + g[1] * 0.5 + g[2] * 0.25) / float(sum(self.patch))
        self.calls = []

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        self.calls.append(x.shape[0])
        return np.stack([(c + 1) * np.tanh(0.5 * x[:, 0]) + 0.01 * self.ramp[None] for c in range(self.n_out)], axis=1)


class FakeModel2D:
    def __init__(self, xy, depth):
        self.xy, self.depth = tuple(xy), depth
        self.output_shape = (None,) + self.xy + (1,)
        g = np.meshgrid(*[np.arange(s) for s in self.xy], indexing="ij")
        self.ramp = (g[0] * 1.0 + g[1] * 0.5) / float(sum(self.xy))
        self.calls = []

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        self.calls.append(x.shape[0])
        w = np.arange(1, self.depth + 1, dtype=np.float64)
        w /= w.sum()
        return (np.tanh(0.5 * x) * w).sum(-1, keepdims=True) + 0.01 * self.ramp[None, ..., None]


def test_index_sets(gold):
    meta, arrs = gold
    for c in meta["index_cases"]:
        _, step = T.overlap_and_step(c["patch"], c["pred"], c["overlap_factor"])
        idx = T.patch_indices_full((0, 0, 0), np.subtract(c["vol"], c["patch"]), step)
        assert len(idx) == c["n"], c["name"]
        assert np.array_equal(idx, arrs["idx_" + c["name"]]), c["name"]
    # config 5 -> 36 tiles with starts {0,33,66,96} x {0,65,128} x {0,65,128}
    i5 = arrs["idx_cfg5_f05"]
    assert sorted(set(i5[:, 0])) == [0, 33, 66, 96] and sorted(set(i5[:, 1])) == [0, 65, 128]


def test_overlap_add_volumes(gold):
    meta, arrs = gold
    for c in meta["volume_cases"]:
        rs = np.random.RandomState(c["seed"])
        data = rs.randn(1, *c["vol"])
        if c["kind"] == "3d":
            fm = FakeModel3D(c["patch"], c["n_out"])
        else:
            fm = FakeModel2D(c["patch"][:2], c["patch"][2])
        out = T.patch_wise_prediction(fm, data, c["patch"], c["overlap_factor"], c["batch_size"])
        assert list(out.shape) == c["out_shape"], c["name"]
        assert fm.calls == c["predict_calls"], c["name"]
        np.testing.assert_allclose(out, arrs["out_" + c["name"]], rtol=0, atol=1e-13, err_msg=c["name"])


def test_get_patch_edge_pad(gold):
    meta, arrs = gold
    pc = meta["patch_cases"]
    vol = np.random.RandomState(pc["seed"]).randn(*pc["vol"])
    for k, c in enumerate(pc["cases"]):
        got = T.get_patch(vol, c["shape"], np.array(c["index"]))
        assert np.array_equal(got, arrs["patch_%d" % k])
