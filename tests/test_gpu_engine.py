"""End-to-end GPU parity of the U-Net engine (forward logits, Dice loss, all gradients, Adam trajectory) vs the oracle.

Bars (BASELINE.json north_star): fp32 mode logits <= 1e-3 relative, |dDice| <= 1e-4 vs the CPU restatement on identical
seeded weights / synthetic volumes.  bf16 mode is checked against the same oracle at bf16-appropriate tolerances.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import bar          # noqa: E402  (limit = <= 2x the error measured on MI355X; FMRI_MEASURE=1 records instead of asserting)


def _setup(spatial, depth, base, N, dtype, seed=42):
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    spec = O.Spec((1,) + tuple(spatial), depth=depth, n_base_filters=base)
    W = spec.init_weights(seed)
    rs = np.random.RandomState(7)
    for k in W:  # non-zero biases so that the bias path is exercised
        if k.endswith("/bias"):
            W[k] = (rs.randn(*W[k].shape) * 0.05).astype(np.float32)
    plan = UNetPlan(1, spatial, depth=depth, n_base_filters=base)
    eng = UNetEngine(plan, N, dtype=dtype)
    eng.load_keras_weights(W)
    x, y = O.synthetic_batch((N, 1) + tuple(spatial))
    return spec, W, eng, x, y


def _dev_inputs(eng, x, y):
    xd = torch.from_numpy(x).to("cuda").to(eng.dtype).reshape(x.shape[0], *x.shape[2:], 1).contiguous()
    yd = torch.from_numpy(y).to("cuda").reshape(-1).contiguous()
    return xd, yd


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_cfg1_fp32_forward_backward_adam():
    """BASELINE config 1: depth 3 / 8 base filters, 1x16x64x64 patch, fp32 parity mode."""
    from oracle import unet_oracle as O
    spec, W, eng, x, y = _setup((16, 64, 64), 3, 8, 1, torch.float32)
    xd, yd = _dev_inputs(eng, x, y)
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float64)
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    assert _rel(logits, ref["logits"]) <= 1e-3
    m = eng.metrics_from_sums(sums.cpu().numpy())
    assert abs(m["dice_coefficient"] - ref["dice"]) <= 1e-4
    # gradients of every tensor
    G = eng.G
    for name, L in eng.layout.items():
        gk = ref["grads"][name + "/kernel"]
        if L["kind"] == "conv":
            mine = eng.w_view(name, G).cpu().numpy().reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
        else:
            mine = eng.w_view(name, G).cpu().numpy().T.reshape(gk.shape)
        assert _rel(mine, gk) <= 2e-3, name
        assert _rel(eng.b_view(name, G).cpu().numpy(), ref["grads"][name + "/bias"]) <= 2e-3, name + " bias"
    # three optimizer steps track the oracle's Keras-Adam trajectory
    opt = O.KerasAdam(W, lr=1e-3)
    eng.t = 0
    for step in range(3):
        r = O.train_step(spec, W, opt, x, y)
        s = eng.train_step(xd, yd, 1e-3)
        torch.cuda.synchronize()
        mm = eng.metrics_from_sums(s.cpu().numpy())
        assert abs(mm["loss"] - r["loss"]) <= 1e-4, (step, mm["loss"], r["loss"])
    Wm = eng.export_keras_weights()
    for k in W:
        assert _rel(Wm[k], W[k]) <= 2e-3, k


def test_cfg1_fp32_logits_against_the_committed_oracle_fixture():
    """the same configs[0] forward against tests/golden/oracle_cfg1_golden.npz (committed output of the oracle, make_oracle_fixture.py):
    logits <= 1e-3 relative, Dice <= 1e-4 - the north-star tolerances - without running the oracle"""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cfg1_golden.npz"))
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    spec = O.Spec((1, 16, 64, 64), depth=3, n_base_filters=8)
    W = spec.init_weights(42)                                  # the fixture's weights: glorot_uniform seed 42, zero biases
    eng = UNetEngine(UNetPlan(1, (16, 64, 64), depth=3, n_base_filters=8), 1, dtype=torch.float32)
    eng.load_keras_weights(W)
    x, y = O.synthetic_batch((1, 1, 16, 64, 64))               # input data only (seeds 1234 / 1235); the oracle's forward is not run
    assert float(x.sum()) == float(gold["x_sum"]) and int(y.sum()) == int(gold["y_sum"])
    np.testing.assert_array_equal(W[sorted(W)[0]].astype(np.float32).ravel()[:16], gold["w_first"])
    xd, yd = _dev_inputs(eng, x, y)
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(gold["logits"].shape)
    assert _rel(logits, gold["logits"]) <= 1e-3
    assert abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - float(gold["dice"])) <= 1e-4


def test_depth4_base32_bf16_small_patch():
    """BASELINE config-2 topology (depth 4 / 32 filters, MFMA kernels) on a small 1x8x16x32... patch vs the oracle."""
    from oracle import unet_oracle as O
    spec, W, eng, x, y = _setup((32, 64, 128), 4, 32, 1, torch.bfloat16)
    xd, yd = _dev_inputs(eng, x, y)
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float32)
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    bar("d4b32_small.logits_rel", _rel(logits, ref["logits"]), 1.7e-2)      # measured 8.5e-3;      # bf16 storage of 14 stacked conv outputs
    m = eng.metrics_from_sums(sums.cpu().numpy())
    bar("d4b32_small.dice_abs", abs(m["dice_coefficient"] - ref["dice"]), 1e-5)      # measured 4.2e-6
    worst = 0.0
    for name, L in eng.layout.items():
        gk = ref["grads"][name + "/kernel"]
        if L["kind"] == "conv":
            mine = eng.w_view(name, eng.G).cpu().numpy().reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
        else:
            mine = eng.w_view(name, eng.G).cpu().numpy().T.reshape(gk.shape)
        e = np.linalg.norm(mine.astype(np.float64) - gk) / (np.linalg.norm(gk) + 1e-30)
        worst = max(worst, e)
    bar("d4b32_small.grad_l2_rel", worst, 4.8e-2)                                     # measured 2.4e-2


def test_bf16_matches_fp32_engine_on_gpu():
    """the two HIP paths (generic fp32 vs MFMA bf16) agree with each other on the same inputs"""
    spec, W, e32, x, y = _setup((16, 32, 32), 3, 32, 2, torch.float32)
    from fmri_hip.engine import UNetEngine
    e16 = UNetEngine(e32.plan, 2, dtype=torch.bfloat16)
    e16.load_keras_weights(W)
    x32, yd = _dev_inputs(e32, x, y)
    x16, _ = _dev_inputs(e16, x, y)
    e32.forward(x32)
    e16.forward(x16)
    torch.cuda.synchronize()
    bar("bf16_vs_fp32_engine.logits_rel", _rel(e16.logits.cpu().numpy(), e32.logits.cpu().numpy()), 1.3e-2)      # measured 6.2e-3


def test_training_loss_decreases_bf16():
    spec, W, eng, x, y = _setup((16, 32, 32), 3, 32, 2, torch.bfloat16)
    xd, yd = _dev_inputs(eng, x, y)
    losses = []
    for _ in range(12):
        s = eng.train_step(xd, yd, 1e-3)
        losses.append(eng.metrics_from_sums(s.cpu().numpy())["loss"])
    assert all(np.isfinite(losses)), losses
    # 12 Adam steps on ONE batch: measured drop 0.05-0.07 on MI355X over repeated runs (atomic-order weight gradients move it in the 4th digit)
    bar("train_bf16.loss_rise_over_12_steps", losses[-1] - losses[0], -0.02)


def test_unet2d_fp32_and_bf16_vs_oracle():
    """2-D twin (reference model/unet/unet.py:22-88): channels-last (N,X,Y,C) slices, planar kernels."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    N, X, Y, C = 4, 32, 64, 5
    for dtype, base, tol_l, tol_d, tol_g in ((torch.float32, 8, 1e-5, 1e-7, 1e-5), (torch.bfloat16, 32, 2.2e-2, 2.2e-5, 1e-1)):   # bf16: first-layer gradient passes through 13 bf16-stored tensors
        spec = O.Spec((X, Y, C), ndim=2, depth=3, n_base_filters=base)
        W = spec.init_weights(11)
        rs = np.random.RandomState(3)
        x = rs.randn(N, X, Y, C).astype(np.float32)
        y = (rs.rand(N, X, Y, 1) > 0.7).astype(np.uint8)
        ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float64 if dtype == torch.float32 else torch.float32)
        eng = UNetEngine(UNetPlan(C, (X, Y), depth=3, n_base_filters=base, ndim=2), N, dtype=dtype)
        eng.load_keras_weights(W)
        xd = torch.from_numpy(x).cuda().to(dtype).unsqueeze(0).contiguous()
        yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
        eng.forward(xd)
        sums = eng.loss_forward(yd)
        eng.backward(yd)
        torch.cuda.synchronize()
        logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
        tag = "unet2d.%s." % ("f32" if dtype == torch.float32 else "bf16")
        bar(tag + "logits_rel", _rel(logits, ref["logits"]), tol_l)
        bar(tag + "dice_abs", abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]), tol_d)
        Wg = {}
        for name, L in eng.layout.items():
            gk = ref["grads"][name + "/kernel"]
            if L["kind"] == "conv":
                mine = eng.w_view(name, eng.G).cpu().numpy().reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
                assert float(np.abs(mine[0]).max()) == 0 and float(np.abs(mine[2]).max()) == 0     # dead kd planes stay untouched
                mine = mine[1]
            else:
                mine = eng.w_view(name, eng.G).cpu().numpy().T.reshape(gk.shape)
            e = np.linalg.norm(mine.astype(np.float64) - gk) / (np.linalg.norm(gk) + 1e-30)
            bar(tag + "grad_l2_rel", e, tol_g)
        Wx = eng.export_keras_weights()
        for k in W:
            assert Wx[k].shape == W[k].shape


def test_unet2d_full_resolution_bf16_vs_oracle():
    """BASELINE configs[3] at its own resolution against the CPU ORACLE (the oracle had seen the 2-D model at toy size
    only, the full-size check compared HIP bf16 with HIP fp32): 8 slices of 256x256x5, depth 4 / 32 filters, bf16 on the planar MFMA
    kernels (first-layer MFMA kernel with (tap, channel) pairs as k, planar parity form in the decoder: 8 slices tile) vs
    oracle.loss_and_grads in fp32 - logits, Dice, every parameter gradient (reference model/unet/unet.py:22-88)."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    N, X, Y, C = 8, 256, 256, 5
    spec = O.Spec((X, Y, C), ndim=2, depth=4, n_base_filters=32)
    W = spec.init_weights(11)
    rs = np.random.RandomState(3)
    x = rs.randn(N, X, Y, C).astype(np.float32)
    y = (rs.rand(N, X, Y, 1) > 0.7).astype(np.uint8)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float32)
    eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), N, dtype=torch.bfloat16)
    assert len(eng.Wup) == 3 and all(eng._use_upcat(n) for n in eng.Wup)          # the planar parity form is what runs
    eng.load_keras_weights(W)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).unsqueeze(0).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    bar("unet2d_full.bf16.logits_rel", _rel(logits, ref["logits"]), 1.5e-2)                 # measured 7.3e-3
    bar("unet2d_full.bf16.dice_abs", abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]), 4.7e-5)   # measured 2.3e-5
    worst = 0.0
    for name, L in eng.layout.items():
        gk = ref["grads"][name + "/kernel"]
        if L["kind"] == "conv":
            mine = eng.w_view(name, eng.G).cpu().numpy().reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
            assert float(np.abs(mine[0]).max()) == 0 and float(np.abs(mine[2]).max()) == 0     # dead kd planes stay untouched
            mine = mine[1]
        else:
            mine = eng.w_view(name, eng.G).cpu().numpy().T.reshape(gk.shape)
        worst = max(worst, float(np.linalg.norm(mine.astype(np.float64) - gk) / (np.linalg.norm(gk) + 1e-30)))
    bar("unet2d_full.bf16.worst_grad_l2_rel", worst, 2.6e-2)                                # measured 1.28e-2


@pytest.mark.parametrize("slices", [8, 6])
def test_unet2d_parity_form_equals_9_tap_kernels(monkeypatch, slices):
    """2-D decoder (UpSampling2D -> concatenate -> Conv2D, reference model/unet/unet.py:60-66): the parity form (4 classes x 2x2 taps on
    the low-res slices, fmri_conv2d_upcat_*) against the engine that keeps the fused-upsample 9-tap kernels (FMRI_UPCAT=0) - same weights,
    same batch: logits and every parameter gradient.  6 slices do not tile (4 per workgroup tile): the engine must fall back by itself."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    X, Y, C = 32, 64, 3
    plan = lambda: UNetPlan(C, (X, Y), depth=3, n_base_filters=32, ndim=2)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(1, slices, X, Y, C).astype(np.float32)).cuda().to(torch.bfloat16)
    y = torch.from_numpy((rs.rand(slices * X * Y) > 0.7).astype(np.uint8)).cuda()
    monkeypatch.setenv("FMRI_UPCAT", "1")
    a = UNetEngine(plan(), slices, dtype=torch.bfloat16, seed=9)
    assert len(a.Wup) == 2 and a._use_upcat(next(iter(a.Wup))) == (slices % 4 == 0)
    monkeypatch.setenv("FMRI_UPCAT", "0")
    b = UNetEngine(plan(), slices, dtype=torch.bfloat16, seed=9)
    assert not b.Wup
    out = []
    for eng in (a, b):
        eng.forward(x)
        eng.loss_forward(y)
        eng.backward(y)
        torch.cuda.synchronize()
        out.append((eng.logits.float().cpu().numpy().copy(), eng.G.cpu().numpy().copy()))
    bar("unet2d_parity_vs_9tap.logits_rel_s%d" % slices, _rel(out[0][0], out[1][0]), (8.2e-3 if slices % 4 == 0 else 0.0))      # measured 4.1e-3
    for name, L in a.layout.items():
        o, n = L["w"]
        ga, gb = out[0][1][o:o + n], out[1][1][o:o + n]
        e = np.linalg.norm(ga - gb) / (np.linalg.norm(gb) + 1e-30)
        assert e <= (6e-2 if slices % 4 == 0 else 1e-6), (name, e)


@pytest.mark.parametrize("norm,deconv", [("batch", True), ("instance", False), ("batch", False), (None, True)])
def test_unet3d_norm_and_deconv_variants_fp32(norm, deconv):
    """the optional pieces of create_convolution_block / get_up_convolution (reference unet.py:103-111,135): BatchNormalization,
    keras-contrib InstanceNormalization, Deconvolution3D — forward, all gradients (incl. gamma/beta and the transposed kernels)."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    spatial, N = (8, 16, 16), 2
    spec = O.Spec((1,) + spatial, depth=2, n_base_filters=8, deconvolution=deconv, batch_normalization=(norm == "batch"),
                  instance_normalization=(norm == "instance"))
    x, y = O.synthetic_batch((N, 1) + spatial)
    # A pre-activation within fp32 rounding of 0 flips its ReLU derivative between the fp32 kernels and the fp64 checker, and with
    # normalisation that single flip moves a whole channel's mean gradient: pick a weight seed without such ties.
    for seed in range(21, 40):
        W = spec.init_weights(seed)
        rs = np.random.RandomState(5)
        for k in W:
            if k.endswith(("/bias", "/beta")):
                W[k] = (rs.randn(*W[k].shape) * 0.05).astype(np.float32)
            if k.endswith("/gamma"):
                W[k] = (1.0 + rs.randn(*W[k].shape) * 0.1).astype(np.float32)
        _, _, inter = O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(x, dtype=torch.float64), return_intermediates=True)
        if min(float(v.abs().min()) for k, v in inter.items() if k.endswith("/z")) > 2e-5:
            break
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float64)
    eng = UNetEngine(UNetPlan(1, spatial, depth=2, n_base_filters=8, norm=norm, deconvolution=deconv), N, dtype=torch.float32)
    eng.load_keras_weights(W)
    xd, yd = _dev_inputs(eng, x, y)
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    assert _rel(logits, ref["logits"]) <= 1e-3
    assert abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]) <= 1e-4
    for name, L in eng.layout.items():
        gk = ref["grads"][name + "/kernel"]
        if L["kind"] == "conv":
            mine = eng.w_view(name, eng.G).cpu().numpy().reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
        elif L["kind"] == "deconv":
            mine = eng.w_view(name, eng.G).cpu().numpy().reshape(2, 2, 2, L["cout"], L["cin"])
        else:
            mine = eng.w_view(name, eng.G).cpu().numpy().T.reshape(gk.shape)
        assert _rel(mine, gk) <= 3e-3, name
        if not (L.get("norm") and norm is not None):
            # with a normalisation layer behind it the conv bias has an exactly-zero gradient: compare absolutely
            assert _rel(eng.b_view(name, eng.G).cpu().numpy(), ref["grads"][name + "/bias"]) <= 3e-3, name + " bias"
        else:
            assert float(np.abs(eng.b_view(name, eng.G).cpu().numpy()).max()) <= 1e-6 * float(np.abs(gk).max() + 1)
            assert _rel(eng.gb_view(name, "gamma", eng.G).cpu().numpy(), ref["grads"][L["norm"] + "/gamma"]) <= 3e-3, name + " gamma"
            assert _rel(eng.gb_view(name, "beta", eng.G).cpu().numpy(), ref["grads"][L["norm"] + "/beta"]) <= 3e-3, name + " beta"
    # a few optimizer steps reduce the loss; weights round-trip through the Keras naming
    l0 = eng.metrics_from_sums(eng.train_step(xd, yd, 1e-2).cpu().numpy())["loss"]
    for _ in range(6):
        s = eng.train_step(xd, yd, 1e-2)
    assert eng.metrics_from_sums(s.cpu().numpy())["loss"] < l0
    Wx = eng.export_keras_weights()
    for k in W:
        assert Wx[k].shape == W[k].shape, k
    eng.predict(xd)                                       # inference path (batch norm: moving averages)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.probs).all()


def _tie_free_weights(init, fwd_z, lo=21, hi=60, thr=2e-5):
    for seed in range(lo, hi):
        W = init(seed)
        if fwd_z(W) > thr:
            return W
    return W


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_norm_tails_match_the_separate_passes_bf16(norm, monkeypatch):
    """bf16 engine with normalisation layers at a size whose level-0 launches have more tiles than CUs (2 x 32x64x64, depth 3, 32
    filters): the statistics summed in the conv epilogues and the backward reductions formed by the input-gradient launches (default)
    against the separate reduction passes (FMRI_NORM_FUSE=0).  The conv outputs are the same bits either way; the sums differ by the order
    of summation, which can move a normalised bf16 value by one unit in the last place."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    spatial, N = (32, 64, 64), 2
    x, y = O.synthetic_batch((N, 1) + spatial)
    out = []
    monkeypatch.setenv("FMRI_NORM_FUSE_MAXLEVEL", "9")
    for fuse in ("3", "0"):                               # 3: the statistics tails and the backward tails (the latter are off by default)
        monkeypatch.setenv("FMRI_NORM_FUSE", fuse)
        eng = UNetEngine(UNetPlan(1, spatial, depth=3, n_base_filters=32, norm=norm), N, dtype=torch.bfloat16, seed=7)
        xd, yd = _dev_inputs(eng, x, y)
        eng.forward(xd)
        eng.loss_forward(yd)
        eng.backward(yd)
        torch.cuda.synchronize()
        fused = sorted(k for k, v in eng._ntail_cache.items() if v) if fuse == "3" else []
        out.append((eng.logits.double().cpu(), eng.G.double().cpu(), fused))
    assert len(out[0][2]) >= 3, out[0][2]             # (32,0,64): enc0b forward; (64,0,64): dec0a skip launch, dec0b forward, dec0b input gradient; (64,0,32): enc0b input gradient
    el = float((out[0][0] - out[1][0]).norm() / out[1][0].norm())
    eg = float((out[0][1] - out[1][1]).norm() / out[1][1].norm())
    print("%s: fused launches %s; logits rel L2 %.2e, gradient buffer rel L2 %.2e" % (norm, out[0][2], el, eg))
    assert el <= 5e-3 and eg <= 2e-2


def test_isensee_graph_engine_fp32_vs_oracle():
    """reference isensee2017.py topology (depth 3, 2 segmentation levels, SpatialDropout3D with fixed masks) on the generic
    layer-graph engine: logits, Dice and every gradient vs the torch-CPU restatement."""
    import fetal_net.model as fmodel
    from fmri_hip.graph_engine import LayerGraphEngine
    from oracle import isensee_oracle as I, unet_oracle as O
    N, sp = 2, (16, 16, 16)
    kw = dict(input_shape=(1,) + sp, depth=3, n_base_filters=4, n_segmentation_levels=2, dropout_rate=0.3)
    model = fmodel.isensee2017_model_3d(**kw)
    spec = I.IsenseeSpec(**kw)
    x, y = O.synthetic_batch((N, 1) + sp)
    rs = np.random.RandomState(8)
    masks = {lv: ((rs.rand(N, spec.levels[lv]["filters"]) < 0.7).astype(np.float64) / 0.7) for lv in range(3)}

    def perturbed(seed):
        W = spec.init_weights(seed)
        r2 = np.random.RandomState(5)
        for k in W:
            if k.endswith(("/bias", "/beta")):
                W[k] = (r2.randn(*W[k].shape) * 0.05).astype(np.float32)
            if k.endswith("/gamma"):
                W[k] = (1.0 + r2.randn(*W[k].shape) * 0.1).astype(np.float32)
        return W

    ref = None
    for seed in range(21, 60):       # avoid LeakyReLU-boundary ties (see test_unet3d_norm_and_deconv_variants_fp32)
        W = perturbed(seed)
        ref = I.loss_and_grads(spec, W, x, y, dropout_masks=masks)
        break
    eng = LayerGraphEngine(model.layers, N, dtype=torch.float32)
    eng.load_keras_weights(W)
    eng.set_dropout_masks({"spatial_dropout3d_%d" % (lv + 1): torch.tensor(masks[lv], dtype=torch.float32).cuda() for lv in range(3)})
    xd = torch.from_numpy(x).cuda().reshape(N, *sp, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    assert _rel(logits, ref["logits"]) <= 1e-3
    assert abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]) <= 1e-4
    Gx = {}
    for name, L in eng.layout.items():
        if L["kind"] == "conv":
            mine = eng.w_view(name, eng.G).cpu().numpy().reshape((L["k"],) * 3 + (L["cout"], L["cin"])).transpose(0, 1, 2, 4, 3)
            gk = ref["grads"][name + "/kernel"]
            e = np.linalg.norm(mine - gk) / (np.linalg.norm(gk) + 1e-30)
            assert e <= 5e-3, (name, e)
        else:
            for key in ("gamma", "beta"):
                gk = ref["grads"][name + "/" + key]
                mine = eng._v(name, key, eng.G).cpu().numpy()
                e = np.linalg.norm(mine - gk) / (np.linalg.norm(gk) + 1e-30)
                assert e <= 5e-3, (name, key, e)
    # training through the public Model surface (random dropout masks) reduces the loss; inference path runs
    eng.set_dropout_masks(None)
    torch.manual_seed(0)                                  # the dropout masks are drawn with torch's device RNG
    losses = [eng.metrics_from_sums(eng.train_step(xd, yd, 5e-3).cpu().numpy())["loss"] for _ in range(20)]
    assert min(losses[-5:]) < losses[0], losses
    eng.predict(xd)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.probs).all()


def test_graph_engine_matches_unet_engine():
    """the generic interpreter and the hand-scheduled engine agree on unet_model_3d (same kernels, different scheduling)"""
    import fetal_net.model as fmodel
    from fmri_hip.engine import UNetEngine, UNetPlan
    from fmri_hip.graph_engine import LayerGraphEngine
    from oracle import unet_oracle as O
    sp, N = (8, 16, 16), 2
    model = fmodel.unet_model_3d(input_shape=(1,) + sp, depth=2, n_base_filters=8)
    W = O.Spec((1,) + sp, depth=2, n_base_filters=8).init_weights(4)
    ge = LayerGraphEngine(model.layers, N, dtype=torch.float32)
    ue = UNetEngine(UNetPlan(1, sp, depth=2, n_base_filters=8), N, dtype=torch.float32)
    ge.load_keras_weights(W)
    ue.load_keras_weights(W)
    x, y = O.synthetic_batch((N, 1) + sp)
    xd = torch.from_numpy(x).cuda().reshape(N, *sp, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    for e in (ge, ue):
        e.forward(xd)
        e.loss_forward(yd)
        e.backward(yd)
    torch.cuda.synchronize()
    assert _rel(ge.logits.cpu().numpy(), ue.logits.cpu().numpy()) <= 1e-5
    for name in ("conv3d_1", "conv3d_3", "conv3d_5", "conv3d_7"):
        a = ge.w_view(name, ge.G).cpu().numpy().reshape(-1)
        b = ue.w_view(name, ue.G).cpu().numpy().reshape(-1)
        assert _rel(a, b) <= 1e-3, name


def test_isensee_bf16_padded_engine(monkeypatch):
    """bf16 mode of the layer-graph engine pads every channel count to 64 and sends all convolutions (stride 2 and 1x1x1 included)
    through the MFMA kernels.  Same weights, inputs and dropout masks on three engines: exact-size fp32 (VALU kernels), exact-size bf16
    (FMRI_GRAPH_PAD=0, VALU kernels) and padded bf16.  Both bf16 engines sit within the bf16 noise of this deep instance-normalised network from fp32
    (measured: 12 % on the first convolutions' weight gradients for either, 8 % between the two - independent rounding), logits and
    Dice agree closely, and the padding channels stay exactly zero."""
    import fetal_net.model as fmodel
    from fmri_hip.graph_engine import LayerGraphEngine
    from oracle import isensee_oracle as I, unet_oracle as O
    N, sp = 2, (16, 32, 32)
    kw = dict(input_shape=(1,) + sp, depth=3, n_base_filters=8, n_segmentation_levels=2, dropout_rate=0.3)
    model = fmodel.isensee2017_model_3d(**kw)
    spec = I.IsenseeSpec(**kw)
    W = spec.init_weights(31)
    r2 = np.random.RandomState(5)
    for k in W:
        if k.endswith(("/bias", "/beta")):
            W[k] = (r2.randn(*W[k].shape) * 0.05).astype(np.float32)
        if k.endswith("/gamma"):
            W[k] = (1.0 + r2.randn(*W[k].shape) * 0.1).astype(np.float32)
    x, y = O.synthetic_batch((N, 1) + sp)
    rs = np.random.RandomState(8)
    masks = {"spatial_dropout3d_%d" % (lv + 1): torch.tensor((rs.rand(N, spec.levels[lv]["filters"]) < 0.7).astype(np.float32) / 0.7).cuda()
             for lv in range(3)}
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    res = {}
    for tag, dt_, pad in (("f32", torch.float32, "1"), ("pad", torch.bfloat16, "1"), ("nopad", torch.bfloat16, "0")):
        monkeypatch.setenv("FMRI_GRAPH_PAD", pad)
        eng = LayerGraphEngine(model.layers, N, dtype=dt_)
        assert eng.pad == (tag == "pad")
        eng.load_keras_weights(W)
        eng.set_dropout_masks(masks)
        xd = torch.from_numpy(x).cuda().reshape(N, *sp, 1).to(dt_).contiguous()
        eng.forward(xd)
        sums = eng.loss_forward(yd)
        eng.backward(yd)
        torch.cuda.synchronize()
        res[tag] = (eng.logits.cpu().numpy().copy(), eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"], eng.G.cpu().numpy().copy(), eng)
    lf, df, gf, ef = res["f32"]
    lb, db, gb, eb = res["pad"]
    ln, dn, gn, _ = res["nopad"]
    rng = np.abs(lf).max()
    bar("isensee_bf16.logits_rel_vs_f32", np.abs(lb - lf).max() / rng, 3e-2)
    bar("isensee_bf16.logits_rel_pad_vs_nopad", np.abs(lb - ln).max() / rng, 1.9e-2)      # measured 9.2e-3
    bar("isensee_bf16.dice_abs", max(abs(db - df), abs(db - dn)), 4e-5)                  # measured 1.8e-5
    normed = {op["name"] for op in ef.ops if op["kind"] == "conv" and any(o2["kind"] == "norm" and o2["ins"][0] == op["out"] for o2 in ef.ops)}
    for name, L in ef.layout.items():
        keys = ("w", "b") if L["kind"] == "conv" else ("gamma", "beta")
        for key in keys:
            if key == "b" and name in normed:
                continue          # the bias of a conv that feeds a normalisation has an exactly zero gradient: only rounding noise to compare
            o, n = L[key]
            ref = np.linalg.norm(gf[o:o + n]) + 1e-30
            bar("isensee_bf16.grad_l2_rel_vs_f32", np.linalg.norm(gb[o:o + n] - gf[o:o + n]) / ref, 0.35)
            bar("isensee_bf16.grad_l2_rel_pad_vs_nopad", np.linalg.norm(gb[o:o + n] - gn[o:o + n]) / ref, 0.27)      # measured 0.13
    # the padding never leaks: channels beyond the logical count are exactly zero in every activation
    for name, t in eb.T.items():
        c = eb.clog[name]
        if t.shape[-1] > c:
            assert float(t[..., c:].abs().max()) == 0.0, name
    # and a few optimiser steps reduce the loss
    eb.set_dropout_masks(None)
    torch.manual_seed(0)
    xd = torch.from_numpy(x).cuda().reshape(N, *sp, 1).to(torch.bfloat16).contiguous()
    losses = [eb.metrics_from_sums(eb.train_step(xd, yd, 5e-3).cpu().numpy())["loss"] for _ in range(12)]
    assert min(losses[-4:]) < losses[0], losses


@pytest.mark.parametrize("fold", ["1", "0"])
def test_deconvolution_variant_bf16_mfma_path_vs_fp32(monkeypatch, fold):
    """Deconvolution3D(k=2,s=2) up-convolution (reference unet.py:132-138 with deconvolution=True) in bf16 against the fp32 engine (VALU
    transposed-conv kernels, the three layers one after the other) on the same weights and batch - with NON-ZERO biases, so that the
    transposed conv's bias on the volume's faces / edges / corners is exercised.  fold = 1 (default): transposed conv + concatenate + conv
    folded into ONE parity-form convolution of the low-res tensor (pre-multiplied filters, per-border-class bias, weight gradients chained
    through the transposed conv's weights); fold = 0: the transposed conv as one-tap parity form, then the plain 27-tap conv."""
    monkeypatch.setenv("FMRI_DECONV_FOLD", fold)
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    sp, N = (16, 32, 64), 2
    res = {}
    x, y = O.synthetic_batch((N, 1) + sp)
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    for dt_ in (torch.float32, torch.bfloat16):
        eng = UNetEngine(UNetPlan(1, sp, depth=2, n_base_filters=32, deconvolution=True), N, dtype=dt_, seed=7)
        assert bool(eng.Wdc) == (dt_ == torch.bfloat16)
        assert bool(eng.Wfd) == (dt_ == torch.bfloat16 and fold == "1")
        g = torch.Generator().manual_seed(21)
        for name, L in eng.layout.items():                         # Keras initialises biases to zero: give every layer a real one
            o, n = L["b"]
            eng.P[o:o + n] = (torch.rand(n, generator=g) - 0.5).cuda() * 0.4
        eng.refresh_weight_copies()
        xd = torch.from_numpy(x).cuda().reshape(N, *sp, 1).to(dt_).contiguous()
        eng.forward(xd)
        eng.loss_forward(yd)
        eng.backward(yd)
        torch.cuda.synchronize()
        res[dt_] = (eng.logits.cpu().numpy().copy(), eng.G.cpu().numpy().copy(), eng)
    lf, gf, ef = res[torch.float32]
    lb, gb, _ = res[torch.bfloat16]
    bar("deconv3d_bf16.logits_rel", np.abs(lb - lf).max() / np.abs(lf).max(), 1.6e-2)
    for name, L in ef.layout.items():
        for key in ("w", "b"):
            o, n = L[key]
            e = np.linalg.norm(gb[o:o + n] - gf[o:o + n]) / (np.linalg.norm(gf[o:o + n]) + 1e-30)
            bar("deconv3d_bf16.grad_l2_rel[%s]" % ("fold" if fold == "1" else "two-step"), e, 4.2e-2)


def test_folded_transposed_conv_border_bias_on_the_device():
    """fmri_conv3d_upcat_fwd_bias27 + fmri_border_class_sums against torch on the CPU: a conv whose bias differs on the volume's faces, edges
    and corners (the classes differ per output voxel), and the per-class sums of a gradient tensor."""
    from fmri_hip import ops
    from fmri_hip.deconv_fold import DeconvFold
    torch.manual_seed(2)
    N, C0, C1, Cout, D, H, W = 1, 32, 32, 64, 8, 16, 32
    bf = torch.bfloat16
    x_low = torch.randn(N, D // 2, H // 2, W // 2, C0).to(bf)
    skip = torch.randn(N, D, H, W, C1).to(bf)
    w3 = torch.randn(27, Cout, C0 + C1) * 0.05
    wt = torch.randn(8, C0, C0) * 0.1
    b3, bt = torch.randn(Cout) * 0.3, torch.randn(C0) * 0.3
    fold = DeconvFold("cuda")
    weff, b27 = fold.effective(w3.cuda(), wt.cuda(), b3.cuda(), bt.cuda(), C0)
    y = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_upcat_fwd_bias27(x_low.cuda(), skip.cuda(), weff.to(bf), w3[:, :, C0:].to(bf).cuda().contiguous(), b27, y, act=0)
    torch.cuda.synchronize()
    # reference: the three layers in fp64 on the bf16-rounded tensors, weights as the kernels see them is NOT what we want here - the point
    # is the bias classes: compare against the same folded evaluation done by torch (weights rounded to bf16 like the kernel's)
    import torch.nn.functional as F_
    xl = x_low.double().permute(0, 4, 1, 2, 3)
    sk = skip.double().permute(0, 4, 1, 2, 3)
    kt = wt.to(bf).double().view(2, 2, 2, C0, C0).permute(4, 3, 0, 1, 2)
    up = F_.conv_transpose3d(xl, kt, None, stride=2)
    k3 = w3.to(bf).double().view(3, 3, 3, Cout, C0 + C1).permute(3, 4, 0, 1, 2)
    ref = F_.conv3d(torch.cat([up, sk], 1), k3, None, padding=1)
    cls = lambda n: torch.tensor([0 if i == 0 else (2 if i == n - 1 else 1) for i in range(n)])
    c = (cls(D)[:, None, None] * 3 + cls(H)[None, :, None]) * 3 + cls(W)[None, None, :]
    ref = ref + b27.cpu().double()[c].permute(3, 0, 1, 2)[None]
    got = y.float().cpu().double().permute(0, 4, 1, 2, 3)
    # (the kernel multiplies pre-multiplied bf16 filters, the reference bf16 factors: 2e-2 of the range covers that; a wrong bias class is 0.3)
    err = (got - ref).abs()
    assert float(err.max()) < 2.5e-2 * float(ref.abs().max()), float(err.max() / ref.abs().max())
    face = (c != 13)
    assert float(err[..., face].max()) < 2.5e-2 * float(ref.abs().max())
    # per-class sums
    dy = torch.randn(2, D, H, W, Cout).to(bf)
    out = torch.zeros(27, Cout, device="cuda")
    ops.border_class_sums(dy.cuda(), out)
    torch.cuda.synchronize()
    want = torch.zeros(27, Cout, dtype=torch.float64)
    want.index_add_(0, c.reshape(-1).repeat(2), dy.double().reshape(-1, Cout))
    want[13] = 0
    assert torch.allclose(out.cpu().double(), want, rtol=1e-4, atol=1e-3)


def test_2d_deconvolution_variant_bf16_mfma_path_vs_fp32():
    """Deconvolution2D(k=2,s=2) of the 2-D model in bf16 = one planar centre-tap MFMA conv to 4*Cout channels + depth-to-space copy; same
    weights and batch on the fp32 engine (VALU transposed-conv kernels): logits and every parameter gradient agree to bf16 tolerance."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    S, sp = 8, (32, 64)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, S) + sp + (5,), generator=g)
    y = (torch.rand((S * sp[0] * sp[1],), generator=g) > 0.6).to(torch.uint8).cuda()
    res = {}
    for dt_ in (torch.float32, torch.bfloat16):
        eng = UNetEngine(UNetPlan(5, sp, depth=2, n_base_filters=32, ndim=2, deconvolution=True), S, dtype=dt_, seed=9)
        assert bool(eng.Wd2) == (dt_ == torch.bfloat16)
        eng.forward(x.cuda().to(dt_).contiguous())
        eng.loss_forward(y)
        eng.backward(y)
        torch.cuda.synchronize()
        res[dt_] = (eng.logits.cpu().numpy().copy(), eng.G.cpu().numpy().copy(), eng)
    lf, gf, ef = res[torch.float32]
    lb, gb, _ = res[torch.bfloat16]
    bar("deconv2d_bf16.logits_rel", np.abs(lb - lf).max() / np.abs(lf).max(), 1.7e-2)      # measured 8.1e-3
    for name, L in ef.layout.items():
        for key in ("w", "b"):
            o, n = L[key]
            ref = np.linalg.norm(gf[o:o + n])
            if ref < 1e-12:
                continue                      # the unused half (ad = 1 taps) of the transposed-conv filter in planar mode
            e = np.linalg.norm(gb[o:o + n] - gf[o:o + n]) / ref
            bar("deconv2d_bf16.grad_l2_rel", e, 8e-2)                                            # measured 5.1e-2


@pytest.mark.parametrize("summation", [False, True])
def test_isensee2d_graph_engine_fp32_vs_oracle(summation):
    """2-D Isensee (reference fetal_net/model/unet/isensee.py:14-105: Conv2D blocks with InstanceNormalization + LeakyReLU, stride-2
    in-convs, SpatialDropout2D with fixed masks, UpSampling2D, 1x1 localisation convs and heads; heads summed only with summation=True)
    on the layer-graph engine in PLANAR mode - the batch of slices is one [1][N][X][Y][C] tensor, 2-D filters are the centre plane of
    27-tap images - against the torch-CPU restatement: logits <= 1e-3 relative, Dice <= 1e-4, every parameter gradient; the dead kd
    planes of every 3x3 filter gradient stay exactly zero."""
    import fetal_net.model as fmodel
    from fmri_hip.graph_engine import LayerGraphEngine
    from oracle import isensee_oracle as I
    N, X, Y, C = 4, 32, 32, 3
    kw = dict(input_shape=(X, Y, C), depth=3, n_base_filters=4, n_segmentation_levels=2, dropout_rate=0.3, summation=summation)
    model = fmodel.isensee2017_model(**kw)
    spec = I.IsenseeSpec(ndim=2, **kw)
    rs = np.random.RandomState(12)
    x = rs.randn(N, X, Y, C).astype(np.float32)
    y = (rs.rand(N, X, Y, 1) > 0.7).astype(np.uint8)
    masks = {lv: ((rs.rand(N, spec.levels[lv]["filters"]) < 0.7).astype(np.float64) / 0.7) for lv in range(3)}
    W = spec.init_weights(23)
    r2 = np.random.RandomState(5)
    for k in W:
        if k.endswith(("/bias", "/beta")):
            W[k] = (r2.randn(*W[k].shape) * 0.05).astype(np.float32)
        if k.endswith("/gamma"):
            W[k] = (1.0 + r2.randn(*W[k].shape) * 0.1).astype(np.float32)
    ref = I.loss_and_grads(spec, W, x, y, dropout_masks=masks)
    eng = LayerGraphEngine(model.layers, N, dtype=torch.float32)
    assert eng.planar and eng.nd == 2
    eng.load_keras_weights(W)
    eng.set_dropout_masks({"spatial_dropout2d_%d" % (lv + 1): torch.tensor(masks[lv], dtype=torch.float32).cuda() for lv in range(3)})
    xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy().reshape(ref["logits"].shape)
    assert _rel(logits, ref["logits"]) <= 1e-3
    assert abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]) <= 1e-4
    assert len(eng.layout) * 2 == len(ref["grads"])
    for name, L in eng.layout.items():
        if L["kind"] == "conv":
            g3 = eng.w_view(name, eng.G).cpu().numpy().reshape((L["k"],) * 3 + (L["cout"], L["cin"])).transpose(0, 1, 2, 4, 3)
            if L["k"] == 3:
                assert float(np.abs(g3[0]).max()) == 0.0 and float(np.abs(g3[2]).max()) == 0.0, name
            for key, mine in (("kernel", g3[L["k"] // 2]), ("bias", eng._v(name, "b", eng.G).cpu().numpy())):
                gk = ref["grads"][name + "/" + key]
                if key == "bias" and float(np.abs(gk).max()) < 1e-9:
                    continue                  # the bias of a conv in front of a normalisation: exactly zero gradient, only noise to compare
                e = np.linalg.norm(mine - gk) / (np.linalg.norm(gk) + 1e-30)
                assert e <= 5e-3, (name, key, e)
        else:
            for key in ("gamma", "beta"):
                gk = ref["grads"][name + "/" + key]
                e = np.linalg.norm(eng._v(name, key, eng.G).cpu().numpy() - gk) / (np.linalg.norm(gk) + 1e-30)
                assert e <= 5e-3, (name, key, e)
    # weights survive the Keras round trip in their 2-D shapes
    Wx = eng.export_keras_weights()
    assert set(Wx) == set(W) and all(Wx[k].shape == W[k].shape and np.array_equal(Wx[k], W[k]) for k in W)
    # a few optimiser steps with random dropout reduce the loss; inference runs
    eng.set_dropout_masks(None)
    torch.manual_seed(0)
    losses = [eng.metrics_from_sums(eng.train_step(xd, yd, 5e-3).cpu().numpy())["loss"] for _ in range(20)]
    assert min(losses[-5:]) < losses[0], losses
    eng.predict(xd)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.probs).all()


def test_isensee2d_bf16_engine_vs_fp32_and_model_surface(tmp_path, monkeypatch):
    """2-D Isensee in bf16 (channels padded to 32, every conv - stride 2 and 1x1 included - on the planar MFMA kernels where the shape
    tiles, generic kernels elsewhere) against the fp32 engine on the same weights / slices / dropout masks, then through the Keras-style
    surface: fit_generator on (N,X,Y,C) batches, predict, save and load_old_model from the file alone."""
    import fetal_net.model as fmodel
    from fetal_net.training import load_old_model
    from fmri_hip.graph_engine import LayerGraphEngine
    from oracle import isensee_oracle as I
    N, X, Y, C = 8, 64, 64, 5
    kw = dict(input_shape=(X, Y, C), depth=3, n_base_filters=16, n_segmentation_levels=2, dropout_rate=0.3)
    model = fmodel.isensee2017_model(**kw)
    spec = I.IsenseeSpec(ndim=2, **kw)
    W = spec.init_weights(4)
    rs = np.random.RandomState(2)
    x = rs.randn(N, X, Y, C).astype(np.float32)
    y = (rs.rand(N, X, Y, 1) > 0.7).astype(np.uint8)
    masks = {"spatial_dropout2d_%d" % (lv + 1): torch.tensor((rs.rand(N, spec.levels[lv]["filters"]) < 0.7).astype(np.float32) / 0.7).cuda()
             for lv in range(3)}
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    res = {}
    for dt_ in (torch.float32, torch.bfloat16):
        eng = LayerGraphEngine(model.layers, N, dtype=dt_)
        eng.load_keras_weights(W)
        eng.set_dropout_masks(masks)
        eng.forward(torch.from_numpy(x).cuda().unsqueeze(0).to(dt_).contiguous())
        s = eng.loss_forward(yd).cpu().numpy().copy()
        eng.backward(yd)
        torch.cuda.synchronize()
        res[dt_] = (eng.logits.cpu().numpy().copy(), eng.metrics_from_sums(s)["dice_coefficient"], eng.G.cpu().numpy().copy(), eng.layout)
    lf, df, gf, layout = res[torch.float32]
    lb, db, gb, _ = res[torch.bfloat16]
    bar("isensee2d_bf16.logits_rel_vs_f32", np.abs(lb - lf).max() / np.abs(lf).max(), 3e-2)
    bar("isensee2d_bf16.dice_abs", abs(db - df), 1e-4)
    for name, L in layout.items():
        for key in (("w",) if L["kind"] == "conv" else ("gamma", "beta")):
            o, n = L[key]
            bar("isensee2d_bf16.grad_l2_rel_vs_f32", np.linalg.norm(gb[o:o + n] - gf[o:o + n]) / (np.linalg.norm(gf[o:o + n]) + 1e-30), 0.35)
    # Keras-style surface
    monkeypatch.setenv("FMRI_DTYPE", "bf16")
    m = fmodel.isensee2017_model(**kw)
    m.set_weights_dict(W)

    def gen():
        while True:
            yield x, y

    hist = m.fit_generator(generator=gen(), steps_per_epoch=4, epochs=2, validation_data=gen(), validation_steps=1, verbose=0)
    assert len(hist.history["loss"]) == 2 and np.isfinite(hist.history["val_loss"]).all()
    p = m.predict(x)
    assert p.shape == (N, X, Y, 1) and np.isfinite(p).all() and 0.0 <= p.min() and p.max() <= 1.0
    path = str(tmp_path / "isensee2d-epoch01-loss-0.100-acc0.900.h5")
    m.save(path)
    again = load_old_model(path, verbose=False)
    assert [l.name for l in again.layers] == [l.name for l in m.layers]
    np.testing.assert_allclose(again.predict(x), p, atol=1e-6)
