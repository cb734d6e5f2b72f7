"""Oracle U-Net: topology vs reference-derived golden JSON; torch-CPU ops vs independent numpy ops; Keras Adam."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import numpy_ref as NR
from oracle import unet_oracle as O


@pytest.fixture(scope="module")
def topo(golden_dir):
    with open(os.path.join(golden_dir, "topology_golden.json")) as f:
        return json.load(f)


def _golden_convs(t):
    return [(l["name"], l["input_shapes"][0][1], l["output_shape"][1]) for l in t["layers"]
            if l["class"] in ("Conv3D", "Conv2D", "Deconvolution3D", "Deconvolution2D")]


@pytest.mark.parametrize("case,kw", [
    ("unet3d_cfg1", dict(input_shape=(1, 16, 64, 64), depth=3, n_base_filters=8)),
    ("unet3d_default", dict(input_shape=(1, 64, 128, 128))),
    ("unet3d_test_model", dict(input_shape=(1, 16, 16, 16), depth=2, deconvolution=True, batch_normalization=True)),
    ("unet2d_cfg4", dict(input_shape=(256, 256, 5), ndim=2)),
])
def test_topology_matches_reference(topo, case, kw):
    spec = O.Spec(**kw)
    mine = [(b["name"], b["cin"], b["cout"]) for b in spec.conv_blocks()]
    for dlv in spec.dec:
        if dlv["up"] is not None:
            mine.append((dlv["up"]["name"], dlv["up"]["cin"], dlv["up"]["cout"]))
    mine.append((spec.final["name"], spec.final["cin"], spec.final["cout"]))
    assert sorted(mine) == sorted(_golden_convs(topo[case]))
    # concat order: up first, then skip (reference unet.py:61)
    for l in topo[case]["layers"]:
        if l["class"] == "Concatenate":
            assert l["inputs"][0].startswith(("up_sampling", "conv3d_transpose", "conv2d_transpose"))


def test_param_counts():
    assert O.Spec((1, 16, 64, 64), depth=3, n_base_filters=8).n_params() == 245873
    assert O.Spec((1, 64, 128, 128)).n_params() == 16315585
    assert O.Spec((256, 256, 5), ndim=2).n_params() == 5441281


def test_torch_ops_match_numpy_ops():
    rs = np.random.RandomState(5)
    x = rs.randn(2, 3, 4, 6, 4)
    k = rs.randn(3, 3, 3, 3, 5)
    b = rs.randn(5)
    got = O._conv(torch.tensor(x), torch.tensor(k), torch.tensor(b), 3).numpy()
    np.testing.assert_allclose(got, NR.conv_same(x, k, b), atol=1e-10)
    np.testing.assert_allclose(torch.nn.functional.max_pool3d(torch.tensor(x), 2).numpy(), NR.maxpool2(x))
    np.testing.assert_allclose(O._upsample(torch.tensor(x), 3).numpy(), NR.upsample2(x))
    kt = rs.randn(2, 2, 2, 4, 3)
    np.testing.assert_allclose(O._deconv(torch.tensor(x), torch.tensor(kt), torch.tensor(rs.randn(4) * 0), 3).numpy(),
                               NR.deconv_k2s2(x, kt), atol=1e-10)
    k1 = rs.randn(1, 1, 1, 3, 2)
    np.testing.assert_allclose(O._conv(torch.tensor(x), torch.tensor(k1), None, 3).numpy(), NR.conv_same(x, k1), atol=1e-10)


def test_forward_matches_numpy_chain():
    spec = O.Spec((1, 4, 8, 8), depth=2, n_base_filters=2)
    W = spec.init_weights(3)
    for k in W:
        if k.endswith("bias"):
            W[k] = np.random.RandomState(1).randn(*W[k].shape).astype(np.float32) * 0.1
    x = np.random.RandomState(2).randn(1, 1, 4, 8, 8)
    logits, probs = O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(x))
    # numpy chain
    r = lambda h: np.maximum(h, 0)
    c = lambda h, n: NR.conv_same(h, W[n + "/kernel"], W[n + "/bias"])
    e0 = r(c(r(c(x, "conv3d_1")), "conv3d_2"))
    e1 = r(c(r(c(NR.maxpool2(e0), "conv3d_3")), "conv3d_4"))
    cat = np.concatenate([NR.upsample2(e1), e0], axis=1)
    d0 = r(c(r(c(cat, "conv3d_5")), "conv3d_6"))
    lg = c(d0, "conv3d_7")
    np.testing.assert_allclose(logits.numpy(), lg, atol=1e-10)
    np.testing.assert_allclose(probs.numpy(), NR.sigmoid(lg), atol=1e-10)


def test_dice_gradient_closed_form():
    # dL/dp = -[2y(Sy+Sp+1) - (2I+1)]/(Sy+Sp+1)^2    vs autograd
    rs = np.random.RandomState(0)
    y = torch.tensor((rs.rand(2, 1, 4, 4, 4) > 0.6).astype(np.float64))
    p = torch.tensor(rs.rand(2, 1, 4, 4, 4), requires_grad=True)
    loss = -O.dice_coefficient_t(y, p)
    loss.backward()
    I, Sy, Sp = float((y * p).sum().detach()), float(y.sum()), float(p.sum().detach())
    den = Sy + Sp + 1
    closed = -(2 * y.numpy() * den - (2 * I + 1)) / den ** 2
    np.testing.assert_allclose(p.grad.numpy(), closed, atol=1e-12)


def test_keras_adam_first_step_is_lr_sign():
    W = {"a": np.array([1.0, -2.0, 3.0])}
    opt = O.KerasAdam(W, lr=0.1)
    g = {"a": np.array([0.5, -0.25, 0.0])}
    opt.step(W, g)
    # t=1: m=(1-b1)g, v=(1-b2)g^2, lr_t = lr*sqrt(1-b2)/(1-b1) -> p -= lr * g/(|g| + eps*sqrt(1-b2)...) ~ lr*sign(g)
    np.testing.assert_allclose(W["a"], [1.0 - 0.1, -2.0 + 0.1, 3.0], atol=1e-5)


def test_training_reduces_loss():
    spec = O.Spec((1, 8, 16, 16), depth=2, n_base_filters=4)
    W = spec.init_weights(42)
    x, y = O.synthetic_batch((1, 1, 8, 16, 16))
    opt = O.KerasAdam(W, lr=1e-2)
    losses = [O.train_step(spec, W, opt, x, y)["loss"] for _ in range(8)]
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("case,kw", [
    ("isensee3d_d3", dict(input_shape=(1, 32, 32, 32), depth=3, n_base_filters=8, n_segmentation_levels=2)),
    ("isensee3d_default", dict()),
])
def test_isensee_oracle_topology_matches_reference(topo, case, kw):
    from oracle.isensee_oracle import IsenseeSpec
    spec = IsenseeSpec(**kw)
    gold = topo[case]["layers"]
    convs = [(l["name"], l["input_shapes"][0][1], l["output_shape"][1], tuple(l["args"][1]) if len(l["args"]) > 1 else tuple(l["kw"]["kernel_size"]),
              tuple(l["kw"].get("strides", (1, 1, 1)))) for l in gold if l["class"] == "Conv3D"]
    mine = [(b["name"], b["cin"], b["cout"], (b["k"],) * 3, (b["s"],) * 3) for b in spec.blocks]
    mine += [(h["name"], h["cin"], h["cout"], (1, 1, 1), (1, 1, 1)) for h in spec.heads.values()]
    assert sorted(mine) == sorted(convs)
    # every conv block is followed by InstanceNormalization then LeakyReLU; concat order is [skip, up]
    by_name = {l["name"]: l for l in gold}
    for b in spec.blocks:
        assert by_name[b["norm"]]["inputs"] == [b["name"]]
    for l in gold:
        if l["class"] == "Concatenate":
            assert l["inputs"][0].startswith("add_") and l["inputs"][1].startswith("leaky_re_lu_")


def test_isensee_oracle_runs_and_trains():
    from oracle import isensee_oracle as I
    spec = I.IsenseeSpec(input_shape=(1, 16, 16, 16), depth=3, n_base_filters=4, n_segmentation_levels=2)
    W = spec.init_weights(3)
    x, y = O.synthetic_batch((2, 1, 16, 16, 16))
    r = I.loss_and_grads(spec, W, x, y)
    assert r["logits"].shape == (2, 1, 16, 16, 16) and all(np.isfinite(g).all() for g in r["grads"].values())
    opt = O.KerasAdam(W, lr=5e-3)
    l0 = r["loss"]
    for _ in range(8):
        r = I.loss_and_grads(spec, W, x, y)
        opt.step(W, r["grads"])
    assert r["loss"] < l0


def test_oracle_reproduces_its_committed_cfg1_fixture():
    """tests/golden/oracle_cfg1_golden.npz (made by make_oracle_fixture.py): the seeded configs[0] forward of the oracle - inputs, weights,
    logits and Dice - has not drifted"""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_fixture", os.path.join(here, "make_oracle_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = np.load(os.path.join(here, "oracle_cfg1_golden.npz"))
    now = mod.compute()
    assert float(now["x_sum"]) == float(gold["x_sum"]) and int(now["y_sum"]) == int(gold["y_sum"])
    np.testing.assert_array_equal(now["w_first"], gold["w_first"])
    np.testing.assert_allclose(now["logits"], gold["logits"], rtol=0, atol=1e-6)
    assert abs(float(now["dice"]) - float(gold["dice"])) <= 1e-9
