"""The BENCHMARKED path pinned at its own size: depth 4 / 32 base filters, 64x128x128 patches, bf16, every default
switch of bench.py (warp-specialised forward kernel, parity form of the decoder 'a' convs, weight gradients on their own stream,
XCD-aware tile / column numbering).

 (i)   N = 1: logits, Dice and ALL 30 parameter gradients against oracle.loss_and_grads (fp32, the CPU restatement of
       fetal_net/model/unet3d/unet.py:17-86 + metrics.py:11-32) run live on the box;
 (ii)  N = 4 = the step bench.py times: a live batch of the learnable task against the ORACLE (logits, Dice, all 30 gradient tensors;
       round 5), and the round 1-3 recipe (same seeds) against the fp32 generic-kernel engine at FULL depth (D = 64);
 (iii) exact tests: small dyadic inputs make every product and every partial sum exact in fp32 whatever the summation order, so the
       kernels must reproduce a CPU computation BIT FOR BIT after the one final bf16 rounding - forward, input gradient and weight
       gradient at full size, on shapes spanning several XCD blocks and several weight-gradient column runs;
 (iv)  every bf16 bar below is <= 2x the error measured on MI355X (`FMRI_MEASURE=1 pytest ...` prints the measurements and writes
       gpurun_out/bf16_measured.json instead of asserting).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURE = os.environ.get("FMRI_MEASURE", "0") == "1"
_measured = {}

# bars = 2 x (measured on MI355X, round 2; see profiles/r02_bf16_measured.json), rounded up to 2 digits
BARS = {                                # measured (MI355X, round 2)
    "n1_logits_rel": 1.8e-2,            # 8.9e-3   max |logits - oracle| / max |oracle|  (bf16 storage of 14 stacked conv outputs)
    "n1_dice_abs": 1.2e-5,              # 5.5e-6   (north-star bar for the fp32 mode: 1e-4)
    "n1_grad_l2_rel": 2.6e-2,           # 1.27e-2  worst per-tensor ||g - g_oracle|| / ||g_oracle|| over the 30 tensors
    "n4_logits_rel": 1.2e-2,            # 5.8e-3
    "n4_dice_abs": 5.0e-6,              # 2.3e-6
    "n4_grad_l2_rel": 2.4e-2,           # 1.16e-2
    "n4o_logits_rel": 1.4e-2,           # 6.8e-3   (round 5) batch 4 of the learnable task against the oracle itself
    "n4o_dice_abs": 3.5e-5,             # 1.74e-5
    "n4o_grad_l2_rel": 1.7e-2,          # 8.1e-3
    "cfg3_2d_logits_rel": 1.7e-2,       # 8.2e-3
    "cfg3_2d_dice_abs": 2.2e-5,         # 1.05e-5
    "cfg3_2d_grad_l2_rel": 2.0e-2,      # 1.0e-2
}


def _check(name, value):
    _measured[name] = max(float(value), _measured.get(name, 0.0))
    print("MEASURED %s = %.3e (bar %.1e)" % (name, value, BARS[name]))
    if MEASURE:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "bf16_measured.json"), "w") as f:
            json.dump(_measured, f, indent=1)
        return
    assert value <= BARS[name], (name, value, BARS[name])


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _grad_keras_layout(eng, name):
    """gradient of layer `name` from the engine's flat buffer, in the Keras kernel layout the oracle reports"""
    L = eng.layout[name]
    g = eng.w_view(name, eng.G).cpu().numpy()
    if L["kind"] == "conv":
        return g.reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
    return g.T.reshape(1, 1, 1, L["cin"], L["cout"])


SPATIAL = (64, 128, 128)


def test_n1_full_size_bf16_default_switches_vs_oracle():
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    for k in ("FMRI_UPCAT", "FMRI_FWD_WS", "FMRI_WGRAD_STREAM", "FMRI_WGRAD_SLAB", "FMRI_WGRAD_WS"):
        assert k not in os.environ, "this test pins the DEFAULT switches; unset " + k
    spec = O.Spec((1,) + SPATIAL, depth=4, n_base_filters=32)
    W = spec.init_weights(42)
    rs = np.random.RandomState(7)
    for k in W:
        if k.endswith("/bias"):
            W[k] = (rs.randn(*W[k].shape) * 0.05).astype(np.float32)
    x, y = O.synthetic_batch((1, 1) + SPATIAL)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float32)
    eng = UNetEngine(UNetPlan(1, SPATIAL, depth=4, n_base_filters=32), 1, dtype=torch.bfloat16)
    assert len(eng.upcat) == 3 and eng._wg_stream is not None                     # parity form + weight-gradient stream are on
    eng.load_keras_weights(W)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).reshape(1, *SPATIAL, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    _check("n1_logits_rel", _rel(eng.logits.cpu().numpy().reshape(ref["logits"].shape), ref["logits"]))
    _check("n1_dice_abs", abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]))
    assert len(eng.layout) == 15                                                   # 14 convs + final 1x1x1 = 30 parameter tensors
    worst, worst_b = 0.0, 0.0
    for name in eng.layout:
        e = _l2(_grad_keras_layout(eng, name), ref["grads"][name + "/kernel"])
        eb = _l2(eng.b_view(name, eng.G).cpu().numpy(), ref["grads"][name + "/bias"])
        print("  %-10s kernel %.3e  bias %.3e" % (name, e, eb))
        worst, worst_b = max(worst, e), max(worst_b, eb)
    _check("n1_grad_l2_rel", max(worst, worst_b))


def test_n1_full_size_fp32_on_the_mfma_kernels_vs_oracle():
    """The north-star's own bar - logits within 1e-3 relative, Dice within 1e-4 of the CPU reference path (reference unet3d/unet.py:68,
    metrics.py:11-15) - met by the BENCHMARKED kernel structure (round 6): fp32 mode runs the fp32 instantiation of the
    warp-specialised MFMA kernels (v_mfma_f32_32x32x2_f32: same halo box, LDS-DMA, swizzle, asynchronous drain, pooled-copy tail, parity form
    of the decoder 'a' convs, kd-sharing weight gradient), not the VALU kernels of conv3d_generic.hip.  Every conv of the network except
    the first (Cin = 1) must be on the MFMA path: checked through fmri_conv3d_uses_mfma, and FMRI_F32_MFMA must not be set."""
    from fmri_hip._lib import lib
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    assert MEASURE or "FMRI_F32_MFMA" not in os.environ            # (FMRI_MEASURE=1 FMRI_F32_MFMA=0: the same figures from the VALU kernels, for comparison)
    spec = O.Spec((1,) + SPATIAL, depth=4, n_base_filters=32)
    W = spec.init_weights(42)
    rs = np.random.RandomState(7)
    for k in W:
        if k.endswith("/bias"):
            W[k] = (rs.randn(*W[k].shape) * 0.05).astype(np.float32)
    x, y = O.synthetic_batch((1, 1) + SPATIAL)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float32)
    plan = UNetPlan(1, SPATIAL, depth=4, n_base_filters=32)
    eng = UNetEngine(plan, 1, dtype=torch.float32)
    assert MEASURE or len(eng.upcat) == 3                                          # the parity form of the three decoder 'a' convs, in fp32
    for c in plan.convs_forward_order():
        if c["cin"] == 1 or MEASURE:
            continue
        D, H, Wd = plan.level_dims(c["level"])
        c1 = plan.enc[c["level"]][1]["cout"] if c["name"] in eng.upcat else 0
        assert lib().fmri_conv3d_uses_mfma(c["cin"] - c1, c1, c["cout"], D, H, Wd, 0) & 3 == 3, c["name"]
    eng.load_keras_weights(W)
    xd = torch.from_numpy(x).cuda().reshape(1, *SPATIAL, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    lrel = _rel(eng.logits.cpu().numpy().reshape(ref["logits"].shape), ref["logits"])
    ddice = abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"])
    worst, worst_b = 0.0, 0.0
    for name in eng.layout:
        e = _l2(_grad_keras_layout(eng, name), ref["grads"][name + "/kernel"])
        eb = _l2(eng.b_view(name, eng.G).cpu().numpy(), ref["grads"][name + "/bias"])
        print("  %-10s kernel %.3e  bias %.3e" % (name, e, eb))
        worst, worst_b = max(worst, e), max(worst_b, eb)
    print("MEASURED fp32-on-MFMA at 64x128x128: logits rel %.3e (bar 1e-3), |dDice| %.3e (bar 1e-4), worst gradient tensor l2 rel: kernels %.3e, biases %.3e"
          % (lrel, ddice, worst, worst_b))
    if MEASURE:
        return
    assert lrel <= 1e-3 and ddice <= 1e-4                                           # the north-star bars (BASELINE.json)
    # measured on MI355X (profiles/r06_f32_fullsize_mfma.log): logits 2.8e-6, Dice 3.4e-8, kernels <= 1.6e-4.  The bias gradients of the last two
    # layers are sums of 10^6 terms of both signs that nearly cancel: 2.2e-3 / 1.8e-3 against the fp32 oracle - the SAME figures to four digits
    # from the VALU kernels (profiles/r06_f32_fullsize_valu.log), i.e. the oracle's own fp32 summation, not these kernels.  Bars = 2-3 x measured.
    assert worst <= 4e-4 and worst_b <= 5e-3


def test_n4_live_batch_bf16_default_switches_vs_oracle():
    """the step bench.py times - batch 4 of the learnable task, every default switch - against the ORACLE itself (round 4's review: the N = 4 step
    met the oracle only through the fp32 engine): logits, Dice and all 30 gradient tensors, global-batch Dice over the four samples"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learnable_task as LT
    from fmri_hip.engine import UNetEngine, UNetPlan
    from oracle import unet_oracle as O
    N = 4
    spec = O.Spec((1,) + SPATIAL, depth=4, n_base_filters=32)
    W = spec.init_weights(42)
    rs = np.random.RandomState(8)
    for k in W:
        if k.endswith("/bias"):
            W[k] = (rs.randn(*W[k].shape) * 0.05).astype(np.float32)
    x, y = LT.host_batch(31, N, SPATIAL, dtype=np.float32)                       # (N, 1, X, Y, Z) float32 / uint8
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = O.loss_and_grads(spec, W, x, y, dtype=torch.float32)
    eng = UNetEngine(UNetPlan(1, SPATIAL, depth=4, n_base_filters=32), N, dtype=torch.bfloat16)
    assert len(eng.upcat) == 3 and eng._wg_stream is not None
    eng.load_keras_weights(W)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).reshape(N, *SPATIAL, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    eng.forward(xd)
    sums = eng.loss_forward(yd)
    eng.backward(yd)
    torch.cuda.synchronize()
    _check("n4o_logits_rel", _rel(eng.logits.cpu().numpy().reshape(ref["logits"].shape), ref["logits"]))
    _check("n4o_dice_abs", abs(eng.metrics_from_sums(sums.cpu().numpy())["dice_coefficient"] - ref["dice"]))
    worst = 0.0
    for name in eng.layout:
        worst = max(worst, _l2(_grad_keras_layout(eng, name), ref["grads"][name + "/kernel"]),
                    _l2(eng.b_view(name, eng.G).cpu().numpy(), ref["grads"][name + "/bias"]))
    _check("n4o_grad_l2_rel", worst)


def test_n4_bench_step_bf16_vs_fp32_engine_full_depth():
    """the step bench.py times (batch 4, seeds 1234 / 1235, glorot seed 42) against the fp32 generic-kernel engine on the same data"""
    import bench
    from fmri_hip.engine import UNetEngine, UNetPlan
    N = 4
    x, y = bench.synthetic_batch((N, 1) + SPATIAL, seed_x=1234, seed_y=1235)
    plan = UNetPlan(1, SPATIAL, depth=4, n_base_filters=32)
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        eng = UNetEngine(plan, N, dtype=dtype, seed=42)
        xd = torch.from_numpy(x).cuda().to(dtype).reshape(N, *SPATIAL, 1).contiguous()
        eng.forward(xd)
        sums = eng.loss_forward(yd).cpu().numpy().copy()
        eng.backward(yd)
        torch.cuda.synchronize()
        res[dtype] = dict(logits=eng.logits.cpu().numpy().copy(), dice=eng.metrics_from_sums(sums)["dice_coefficient"],
                          G=eng.G.cpu().numpy().copy(), layout=eng.layout, vox=sums[7])
        del eng
        torch.cuda.empty_cache()
    a, b = res[torch.bfloat16], res[torch.float32]
    assert a["vox"] == b["vox"] == N * np.prod(SPATIAL)
    _check("n4_logits_rel", _rel(a["logits"], b["logits"]))
    _check("n4_dice_abs", abs(a["dice"] - b["dice"]))
    worst = 0.0
    for name, L in a["layout"].items():
        for rng in (L["w"], L["b"]):
            o, n = rng
            worst = max(worst, _l2(a["G"][o:o + n], b["G"][o:o + n]))
    _check("n4_grad_l2_rel", worst)


# ---------------------------------------------------------------------------------------------------------- (iii) exact tests
def _dyadic(shape, lo, hi, den, g, density=1.0):
    """random k/den, k in [lo, hi]; optionally only a `density` fraction non-zero"""
    t = torch.randint(lo, hi + 1, shape, generator=g).float() / den
    if density < 1.0:
        t = t * (torch.rand(shape, generator=g) < density).float()
    return t


def _bf16_bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy()


def _cpu_conv_ndhwc(x, w27, bias=None):
    """fp32 'same' conv on the CPU; x [N,D,H,W,C], w27 [27][Cout][Cin] -> [N,D,H,W,Cout].  Exact for the dyadic data of these tests."""
    Cout, Cin = w27.shape[1], w27.shape[2]
    k = w27.reshape(3, 3, 3, Cout, Cin).permute(3, 4, 0, 1, 2).contiguous()
    y = F.conv3d(x.permute(0, 4, 1, 2, 3).contiguous(), k, bias, padding=1)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def _up2(x):
    for ax in (1, 2, 3):
        x = torch.repeat_interleave(x, 2, dim=ax)
    return x


@pytest.mark.parametrize("case", ["enc0b_n2", "dec0a_27tap", "dec0a_parity"])
def test_forward_is_exact_on_dyadic_data_at_full_size(case):
    """x = k/4 (|x| <= 1), w = k/8 (|w| <= 1/4), bias = k/4: every term is a multiple of 1/32 and the sum of <= 27*192 of them stays
    below 2^24/32, i.e. exact in fp32 in ANY order; the kernel's only rounding is the final bf16 store (RNE = torch's .to(bfloat16)).
    enc0b_n2: 2 x 64x128x128, 32 -> 64 (1,024 tiles: several compact XCD blocks per XCD); dec0a: [up2(128) | 64] -> 64 at 64x128x128
    as fused-upsample 27-tap launch and in parity form (there the up-sampled channels' partial sum is stored as bf16 once more and the
    skip channels' partial sum is rounded to bf16 before the two meet in fp32: the expected value applies those roundings too; every step
    in between is exact, so the comparison stays bit for bit)."""
    from fmri_hip import ops
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(11)
    bf = torch.bfloat16
    D, H, W = SPATIAL
    if case == "enc0b_n2":
        N, C0, C1, Cout = 2, 32, 0, 64
        x0 = _dyadic((N, D, H, W, C0), -4, 4, 4, g)
        x1 = None
    else:
        N, C0, C1, Cout = 1, 128, 64, 64
        x0 = _dyadic((N, D // 2, H // 2, W // 2, C0), -4, 4, 4, g)
        x1 = _dyadic((N, D, H, W, C1), -4, 4, 4, g)
    w = _dyadic((27, Cout, C0 + C1), -2, 2, 8, g)
    bias = _dyadic((Cout,), -4, 4, 4, g)
    y = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    x0d, x1d = x0.to(bf).cuda(), (None if x1 is None else x1.to(bf).cuda())
    if case == "dec0a_parity":
        up_f, sk_f = torch.empty((8, 8, Cout, C0), dtype=bf, device="cuda"), torch.empty((27, Cout, C1), dtype=bf, device="cuda")
        up_d, sk_d = torch.empty((8, 8, C0, Cout), dtype=bf, device="cuda"), torch.empty((27, C1, Cout), dtype=bf, device="cuda")
        ops.conv3d_pack_up_weights(w.cuda(), C0, C1, up_f, up_d, sk_f, sk_d)
        ops.conv3d_upcat_fwd(x0d, x1d, up_f, sk_f, bias.cuda(), y, act=1)
        part_up = _cpu_conv_ndhwc(_up2(x0), w[:, :, :C0].contiguous())            # exact; the kernel stores it as bf16 ...
        part_up = part_up.to(bf).float()
        part_sk = _cpu_conv_ndhwc(x1, w[:, :, C0:].contiguous(), bias).to(bf).float()   # ... the skip part + bias is rounded to bf16 too, and the two meet in fp32
        expect = F.relu(part_up + part_sk)
    else:
        ops.conv3d_fwd(x0d, x1d, w.to(bf).cuda(), bias.cuda(), y, up0=(case != "enc0b_n2"), act=1)
        xin = x0 if x1 is None else torch.cat([_up2(x0), x1], dim=-1)
        expect = F.relu(_cpu_conv_ndhwc(xin, w, bias))
    torch.cuda.synchronize()
    assert float(expect.abs().max()) * 32 < 2 ** 24
    got, exp = _bf16_bits(y), _bf16_bits(expect.to(bf))
    bad = int((got != exp).sum())
    assert bad == 0, "%s: %d of %d outputs differ from the exact value" % (case, bad, got.size)


@pytest.mark.parametrize("case", ["dec0b_n2_masked", "dec0a_parity"])
def test_input_gradient_is_exact_on_dyadic_data_at_full_size(case):
    """dx = conv(dy, flipped W^T) * (mask > 0).  dec0b_n2_masked: plain dgrad 64 -> 64 at 2 x 64x128x128 with the ReLU mask of the producer;
    dec0a_parity: the parity-form input gradient - w.r.t. the LOW-res tensor (= the 2x2x2 sum of the full-resolution gradient, exact here
    because nothing is rounded in between) and w.r.t. the skip tensor."""
    from fmri_hip import ops
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(12)
    bf = torch.bfloat16
    D, H, W = SPATIAL
    if case == "dec0b_n2_masked":
        N, Cin, Cout = 2, 64, 64
        w = _dyadic((27, Cout, Cin), -2, 2, 8, g)
        dy = _dyadic((N, D, H, W, Cout), -4, 4, 4, g)
        mask = _dyadic((N, D, H, W, Cin), -1, 1, 1, g)
        wf, wd = torch.empty((27, Cout, Cin), dtype=bf, device="cuda"), torch.empty((27, Cin, Cout), dtype=bf, device="cuda")
        ops.pack_weights(w.cuda(), wf, wd)
        dx = torch.empty((N, D, H, W, Cin), dtype=bf, device="cuda")
        ops.conv3d_dgrad(dy.to(bf).cuda(), wd, dx, mask=mask.to(bf).cuda())
        # conv-transpose of a 'same' stride-1 conv = 'same' conv with spatially flipped, channel-transposed filters
        wt = w.reshape(3, 3, 3, Cout, Cin).flip(0, 1, 2).permute(0, 1, 2, 4, 3).reshape(27, Cin, Cout).contiguous()
        expect = torch.where(mask > 0, _cpu_conv_ndhwc(dy, wt), torch.zeros(()))      # +0 where masked, as the kernel's bit mask
        torch.cuda.synchronize()
        assert int((_bf16_bits(dx) != _bf16_bits(expect.to(bf))).sum()) == 0
        return
    N, C0, C1, Cout = 1, 128, 64, 64
    w = _dyadic((27, Cout, C0 + C1), -2, 2, 8, g)
    dy = _dyadic((N, D, H, W, Cout), -2, 2, 4, g)
    mask_low = _dyadic((N, D // 2, H // 2, W // 2, C0), -1, 1, 1, g)
    up_d, sk_d = torch.empty((8, 8, C0, Cout), dtype=bf, device="cuda"), torch.empty((27, C1, Cout), dtype=bf, device="cuda")
    up_f, sk_f = torch.empty((8, 8, Cout, C0), dtype=bf, device="cuda"), torch.empty((27, Cout, C1), dtype=bf, device="cuda")
    ops.conv3d_pack_up_weights(w.cuda(), C0, C1, up_f, up_d, sk_f, sk_d)
    dlow = torch.empty((N, D // 2, H // 2, W // 2, C0), dtype=bf, device="cuda")
    dskip = torch.empty((N, D, H, W, C1), dtype=bf, device="cuda")
    ops.conv3d_upcat_dgrad(dy.to(bf).cuda(), up_d, sk_d, mask_low.to(bf).cuda(), None, dlow, dskip)
    wt = w.reshape(3, 3, 3, Cout, C0 + C1).flip(0, 1, 2).permute(0, 1, 2, 4, 3).reshape(27, C0 + C1, Cout).contiguous()
    dcat = _cpu_conv_ndhwc(dy, wt)                                                  # exact full-resolution gradient of the concat
    dup = dcat[..., :C0]
    exp_low = torch.where(mask_low > 0, dup.reshape(N, D // 2, 2, H // 2, 2, W // 2, 2, C0).sum(dim=(2, 4, 6)), torch.zeros(()))
    torch.cuda.synchronize()
    assert float(exp_low.abs().max()) * 32 < 2 ** 24
    assert int((_bf16_bits(dlow) != _bf16_bits(exp_low.to(bf))).sum()) == 0
    assert int((_bf16_bits(dskip) != _bf16_bits(dcat[..., C0:].contiguous().to(bf))).sum()) == 0


@pytest.mark.parametrize("case", ["enc0b_n2", "dec0a_27tap", "dec0a_parity"])
def test_weight_gradient_is_exact_on_dyadic_data_at_full_size(case):
    """dw[tap][co][ci] = sum_v dy[v][co] x[v+tap][ci] over 1-2 M voxels, FULL depth (every column run of every workgroup, both flush
    forms).  dy is sparse (1/16 of the voxels non-zero, k/2) and x = k/4, so every term is a multiple of 1/8 and sum_v |dy| <= 2^17
    bounds every partial sum below 2^24/8: exact in fp32 in any order, atomics included.  The fp32 result must EQUAL the CPU's."""
    from fmri_hip import ops
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(13)
    bf = torch.bfloat16
    D, H, W = SPATIAL
    if case == "enc0b_n2":
        N, C0, C1, Cout = 2, 32, 0, 64
        x0, x1 = _dyadic((N, D, H, W, C0), -4, 4, 4, g), None
    else:
        N, C0, C1, Cout = 1, 128, 64, 64
        x0, x1 = _dyadic((N, D // 2, H // 2, W // 2, C0), -4, 4, 4, g), _dyadic((N, D, H, W, C1), -4, 4, 4, g)
    dy = _dyadic((N, D, H, W, Cout), -2, 2, 2, g, density=1.0 / 16)
    assert float(dy.abs().sum(dim=(0, 1, 2, 3)).max()) * 8 < 2 ** 24               # the exactness bound, per output channel
    Cin = C0 + C1
    dw, db = torch.zeros((27, Cout, Cin), device="cuda"), torch.zeros(Cout, device="cuda")
    x0d, x1d, dyd = x0.to(bf).cuda(), (None if x1 is None else x1.to(bf).cuda()), dy.to(bf).cuda()
    if case == "dec0a_parity":
        ops.conv3d_upcat_wgrad(x0d, x1d, dyd, dw, db, torch.empty(64 * Cout * C0, device="cuda"))
    else:
        ops.conv3d_wgrad(x0d, x1d, dyd, dw, db, up0=(case != "enc0b_n2"))
    xin = x0 if x1 is None else torch.cat([_up2(x0), x1], dim=-1)
    xc = xin.permute(0, 4, 1, 2, 3).contiguous()
    dyc = dy.permute(0, 4, 1, 2, 3).contiguous()
    gw = torch.nn.grad.conv3d_weight(xc, (Cout, Cin, 3, 3, 3), dyc, padding=1)     # [Cout][Cin][3][3][3]
    exp_dw = gw.permute(2, 3, 4, 0, 1).reshape(27, Cout, Cin)
    exp_db = dy.sum(dim=(0, 1, 2, 3))
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), exp_dw), "%s: max |diff| %.3e" % (case, float((dw.cpu() - exp_dw).abs().max()))
    assert torch.equal(db.cpu(), exp_db)


# ------------------------------------------------------------------------------------------- configs[3]: 2-D mode at full size
def test_cfg3_2d_full_size_step_bf16_vs_fp32_engine():
    """BASELINE configs[3]: 64 slices of 256x256x5, depth 4 / 32 filters, one full training step (planar kernels, planar parity form,
    5-channel first layer on the first-layer MFMA kernels) against the fp32 generic-kernel engine: logits, Dice, all gradients; plus
    the size-independent properties: every voxel counted once, sum(y) exact, dead kd planes of the 3x3x3 gradient images untouched."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    S, X, Y, C = 64, 256, 256, 5
    rs = np.random.RandomState(31)
    x = rs.randn(1, S, X, Y, C).astype(np.float32)
    y = (rs.rand(S * X * Y) > 0.7).astype(np.uint8)
    yd = torch.from_numpy(y).cuda()
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), S, dtype=dtype, seed=42)
        xd = torch.from_numpy(x).cuda().to(dtype).contiguous()
        eng.forward(xd)
        sums = eng.loss_forward(yd).cpu().numpy().copy()
        eng.backward(yd)
        torch.cuda.synchronize()
        if dtype == torch.bfloat16:
            assert len(eng.upcat) == 3                                             # the planar parity form is in use
        res[dtype] = dict(logits=eng.logits.cpu().numpy().copy(), sums=sums, dice=eng.metrics_from_sums(sums)["dice_coefficient"],
                          G=eng.G.cpu().numpy().copy(), layout=eng.layout)
        del eng
        torch.cuda.empty_cache()
    a, b = res[torch.bfloat16], res[torch.float32]
    assert a["sums"][7] == b["sums"][7] == S * X * Y and a["sums"][1] == b["sums"][1] == float(y.sum())
    _check("cfg3_2d_logits_rel", _rel(a["logits"], b["logits"]))
    _check("cfg3_2d_dice_abs", abs(a["dice"] - b["dice"]))
    worst = 0.0
    for name, L in a["layout"].items():
        o, n = L["w"]
        if L["kind"] == "conv":
            g27 = a["G"][o:o + n].reshape(3, 9 * L["cout"] * L["cin"])
            assert float(np.abs(g27[0]).max()) == 0.0 and float(np.abs(g27[2]).max()) == 0.0, name
        worst = max(worst, _l2(a["G"][o:o + n], b["G"][o:o + n]))
        ob, nb = L["b"]
        worst = max(worst, _l2(a["G"][ob:ob + nb], b["G"][ob:ob + nb]))
    _check("cfg3_2d_grad_l2_rel", worst)


# ------------------------------------------------------------------------- configs[4]: sliding window over a 160x256x256 volume
@pytest.mark.parametrize("graph", ["1", "0"])
def test_cfg4_full_volume_sliding_window(monkeypatch, graph, golden_dir):
    """BASELINE configs[4]: 160x256x256 volume, 64x128x128 patches, overlap 0.5 -> the 36 tiles of tests/golden/tiler_golden.json (the
    reference's own index list), depth-4 / 32-filter bf16 model, hipGraph replay on / off.  Checks: (a) the tile list; (b) the device
    overlap-add equals the oracle tiler (reference prediction.py:118-210 restated) driven with THE SAME engine as its model, tile by
    tile on the host in float64; (c) a constant model (zero final kernel, bias b) yields sigmoid(b) at every voxel - coverage and count
    normalisation at full size; (d) graph replay == eager, bit for bit."""
    monkeypatch.setenv("FMRI_DTYPE", "bf16")
    monkeypatch.setenv("FMRI_HIPGRAPH", graph)
    import fetal_net.model as fmodel
    from fetal_net.prediction import _geometry, patch_wise_prediction
    from oracle import tiler_oracle
    patch = (64, 128, 128)
    vol = (160, 256, 256)
    model = fmodel.unet_model_3d(input_shape=(1,) + patch, depth=4, n_base_filters=32)
    data = np.random.RandomState(17).randn(1, *vol).astype(np.float32).astype(np.float64)
    ref_idx = np.load(os.path.join(golden_dir, "tiler_golden.npz"))["idx_cfg5_f05"]     # the reference's own index list for this geometry
    _, _, _, _, indices, _ = _geometry(model, data, patch, 0.5)
    assert len(indices) == 36 and np.array_equal(np.asarray(indices), ref_idx)
    out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5, batch_size=5)
    assert out.shape == vol + (1,) and out.dtype == np.float64

    class EngineAsForeignModel:                     # the reference's duck type: .output_shape + .predict(ndarray (B,1,X,Y,Z))
        output_shape = (None, 1) + patch

        def predict(self, xb):
            return model.predict(np.asarray(xb))

    ref = tiler_oracle.patch_wise_prediction(EngineAsForeignModel(), data, patch, 0.5, 5)
    err = float(np.abs(out - ref).max())
    print("cfg4 device overlap-add vs oracle tiler on the same engine: max |diff| %.3e" % err)
    assert err <= 1e-6                              # same bf16 network outputs, fp64 accumulation on both sides; fp32 probabilities
    if graph == "1":
        monkeypatch.setenv("FMRI_HIPGRAPH", "0")
        model.__dict__.pop("_tile_state", None)
        eager = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5, batch_size=5)
        assert np.array_equal(eager, out)
        monkeypatch.setenv("FMRI_HIPGRAPH", "1")
        model.__dict__.pop("_tile_state", None)
    W = model.get_weights_dict()
    last = [k for k in W if k.endswith("/kernel")][-1]
    W[last] = np.zeros_like(W[last])
    W[last.replace("/kernel", "/bias")] = np.full_like(W[last.replace("/kernel", "/bias")], 0.75)
    model.set_weights_dict(W)
    const = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=0.5, batch_size=5)
    s = 1.0 / (1.0 + np.exp(-0.75))
    assert float(np.abs(const - s).max()) <= 1e-6


KD_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join(%r, "fetal-mri-segmentation_amd"))
import numpy as np, torch
from fmri_hip import ops
g = torch.Generator().manual_seed(5)
def dyadic(shape, lo, hi, den, density=1.0):
    t = torch.randint(lo, hi + 1, shape, generator=g).float() / den
    return t * (torch.rand(shape, generator=g) < density).float() if density < 1.0 else t
out = {}
# (a) runs that cross column boundaries (5 units per workgroup on columns of 4 planes), dual source; (b) fused x2 up-sampled source
for tag, (N, D, H, W, C0, C1, Cout, up0) in dict(cols=(5, 4, 128, 128, 64, 0, 64, 0), dual=(1, 8, 32, 64, 64, 64, 128, 0), up=(1, 8, 32, 64, 128, 64, 64, 1)).items():
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    x0, x1 = dyadic(s0, -4, 4, 4), (dyadic((N, D, H, W, C1), -4, 4, 4) if C1 else None)
    dy = dyadic((N, D, H, W, Cout), -2, 2, 2, density=1.0 / 16)
    dw, db = torch.zeros((27, Cout, C0 + C1), device="cuda"), torch.zeros(Cout, device="cuda")
    ops.conv3d_wgrad(x0.to(torch.bfloat16).cuda(), None if x1 is None else x1.to(torch.bfloat16).cuda(), dy.to(torch.bfloat16).cuda(), dw, db, up0=bool(up0))
    torch.cuda.synchronize()
    out[tag + "_dw"], out[tag + "_db"] = dw.cpu().numpy(), db.cpu().numpy()
# (c) the parity-form weight gradient (fmri_conv3d_upcat_wgrad): kd'-sharing kernel (default) against the per-kd' kernel (FMRI_UPW_KD=0)
N, D, H, W, C0, C1, Cout = 2, 8, 32, 64, 128, 64, 64
xl, xs = dyadic((N, D // 2, H // 2, W // 2, C0), -4, 4, 4), dyadic((N, D, H, W, C1), -4, 4, 4)
dy = dyadic((N, D, H, W, Cout), -2, 2, 2, density=1.0 / 16)
dw, db = torch.zeros((27, Cout, C0 + C1), device="cuda"), torch.zeros(Cout, device="cuda")
ops.conv3d_upcat_wgrad(xl.to(torch.bfloat16).cuda(), xs.to(torch.bfloat16).cuda(), dy.to(torch.bfloat16).cuda(), dw, db, torch.empty(64 * Cout * C0, device="cuda"))
torch.cuda.synchronize()
out["upcat_dw"], out["upcat_db"] = dw.cpu().numpy(), db.cpu().numpy()
np.savez(sys.argv[1], **out)
print("DONE")
"""


def test_kd_sharing_weight_gradient_kernels_are_exact(tmp_path):
    """the kd-sharing weight-gradient kernels (one workgroup per (Cout, Cin) block walks columns of d-planes for all 27 taps through a 4-slot
    x-plane ring; FMRI_WGRAD_KD=3 forces them wherever the shape allows: the 4-wave / 32-block form that is the default on every launch without
    fused up-sampling, and
    the 8-wave / 64-block form) against the per-kd kernel (FMRI_WGRAD_KD=0) on dyadic data whose sums are exact in fp32 in any order: all three
    must agree BIT FOR BIT - runs that cross column boundaries, a dual source, a fused up-sampled source; and (round 6) the parity-form weight
    gradient from the kd'-sharing kernel k_conv_wgrad_up_kd (default) against the per-kd' kernel (FMRI_UPW_KD=0)"""
    import subprocess
    import sys
    f = tmp_path / "kd.py"
    f.write_text(KD_SCRIPT % ROOT)
    res = []
    for tag, env in (("kd32", dict(FMRI_WGRAD_KD="3", FMRI_UPW_KD="2")), ("kd64", dict(FMRI_WGRAD_KD="3", FMRI_WGRAD_KD_BLK="64")),      # (kd32 arm: the 8-wave parity form, kd64 arm: the default 4-wave one)
                     ("perkd", dict(FMRI_WGRAD_KD="0", FMRI_UPW_KD="0"))):
        o = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, str(f), o], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and "DONE" in r.stdout, (tag, r.stdout[-1000:], r.stderr[-3000:])
        res.append(np.load(o))
    assert len(res[0].files) == 8
    for k in res[0].files:
        assert float(np.abs(res[2][k]).max()) > 0
        assert np.array_equal(res[0][k], res[2][k]), ("kd32", k)
        assert np.array_equal(res[1][k], res[2][k]), ("kd64", k)


# ---------------------------------------------------------------------------------------------------------- deterministic mode
def test_bench_step_is_bit_reproducible_in_deterministic_mode(monkeypatch):
    """FMRI_DETERMINISTIC=1 at the size bench.py times (depth 4 / 32 filters, 4 x 64x128x128, bf16, two gradient streams): the same weights
    and batch give the SAME BITS in every gradient element and metric sum, pass after pass, and two optimizer steps from the same state end in
    bit-identical parameters.  (The default mode adds workgroups' partial sums with fp32 atomics in arrival order; here they meet as 2^-40
    fixed-point integers in a shadow of the gradient buffer - include/fmri_hip.h, fmri_set_deterministic.)  The deterministic gradients
    agree with the default mode's to fp32 summation-order accuracy."""
    from fmri_hip import ops
    from fmri_hip.engine import UNetEngine, UNetPlan
    import bench
    N = 4
    x, y = bench.synthetic_batch((N, 1) + SPATIAL)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).reshape(N, *SPATIAL, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    plan = UNetPlan(1, SPATIAL, depth=4, n_base_filters=32)
    monkeypatch.setenv("FMRI_DETERMINISTIC", "1")
    eng = UNetEngine(plan, N, dtype=torch.bfloat16, seed=42)
    try:
        assert eng.deterministic and not eng.upcat_wgrad and eng._wg_stream is not None
        P0 = eng.P.clone()
        runs = []
        for _ in range(3):
            eng.forward(xd)
            s = eng.loss_forward(yd).clone()
            eng.backward(yd)
            torch.cuda.synchronize()
            runs.append((eng.G.clone(), s, eng.logits.clone()))
        for G, s, lg in runs[1:]:
            assert torch.equal(G, runs[0][0]), "gradients differ between two passes: %d elements" % int((G != runs[0][0]).sum())
            assert torch.equal(s, runs[0][1]) and torch.equal(lg, runs[0][2])
        assert int(eng.G64.abs().max()) == 0                      # the shadow is cleared by the finish
        assert float(runs[0][0].abs().max()) > 0
        ends = []
        for _ in range(2):
            eng.P.copy_(P0)
            eng.M.zero_()
            eng.V.zero_()
            eng.t = 0
            eng.refresh_weight_copies()
            for _ in range(2):
                eng.train_step(xd, yd, 1e-4)
            torch.cuda.synchronize()
            ends.append(eng.P.clone())
        assert torch.equal(ends[0], ends[1]) and not torch.equal(ends[0], P0)
        # one registration per process: a second deterministic engine is refused while this one lives, accepted after close()
        with pytest.raises(RuntimeError):
            UNetEngine(UNetPlan(1, (8, 16, 32), depth=2, n_base_filters=32), 1, dtype=torch.bfloat16)
    finally:
        eng.close()
    eng2 = UNetEngine(UNetPlan(1, (8, 16, 32), depth=2, n_base_filters=32), 1, dtype=torch.bfloat16)
    assert eng2.deterministic
    del eng2                                                      # garbage collection gives the registration back as well
    import gc
    gc.collect()
    # against the default mode (fp32 atomics, parity-form weight gradient): same gradients up to summation order / the parity form's rounding
    monkeypatch.setenv("FMRI_DETERMINISTIC", "0")
    ref = UNetEngine(plan, N, dtype=torch.bfloat16, seed=42)
    assert not ref.deterministic
    ref.forward(xd)
    ref.loss_forward(yd)
    ref.backward(yd)
    torch.cuda.synchronize()
    assert torch.equal(ref.logits, runs[0][2])
    g0, g1 = runs[0][0].double(), ref.G.double()
    for name, L in ref.layout.items():
        o, n = L["w"]
        e = float((g0[o:o + n] - g1[o:o + n]).norm() / (g1[o:o + n].norm() + 1e-30))
        assert e < (2e-2 if name in ref.upcat_wgrad else 1e-4), (name, e)


def test_folded_transposed_conv_at_full_size_vs_the_two_step_form(monkeypatch):
    """`deconvolution=True` at the benchmark's size (depth 4 / 32 filters, 1 x 64x128x128, bf16): the folded form (one parity-form conv of
    the low-res tensor per decoder level, fmri_hip/deconv_fold.py) against the two-step form (transposed conv materialised, then the plain
    27-tap conv over the concatenation) on the same weights - non-zero biases everywhere - and batch: logits, Dice and every gradient tensor,
    the transposed convs' included.  Both are bf16 paths with different rounding points (the two-step form rounds the up-sampled tensor to
    bf16, the folded one the pre-multiplied filters), so the bars are those of the bf16-vs-fp32 comparisons of this file."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    import bench
    x, y = bench.synthetic_batch((1, 1) + SPATIAL)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).reshape(1, *SPATIAL, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("FMRI_DECONV_FOLD", fold)
        eng = UNetEngine(UNetPlan(1, SPATIAL, depth=4, n_base_filters=32, deconvolution=True), 1, dtype=torch.bfloat16, seed=42)
        assert len(eng.Wfd) == (3 if fold == "1" else 0)
        g = torch.Generator().manual_seed(5)
        for name, L in eng.layout.items():
            o, n = L["b"]
            eng.P[o:o + n] = (torch.rand(n, generator=g) - 0.5).cuda() * 0.2
        eng.refresh_weight_copies()
        eng.forward(xd)
        s = eng.loss_forward(yd).cpu().numpy()
        eng.backward(yd)
        torch.cuda.synchronize()
        res[fold] = (eng.logits.float().cpu().numpy().copy(), eng.metrics_from_sums(s)["dice_coefficient"], eng.G.cpu().numpy().copy(), eng.layout)
        del eng
        torch.cuda.empty_cache()
    la, da, ga, layout = res["1"]
    lb, db_, gb, _ = res["0"]
    e_log = _rel(la, lb)
    worst = 0.0
    for name, L in layout.items():
        for key in ("w", "b"):
            o, n = L[key]
            worst = max(worst, _l2(ga[o:o + n], gb[o:o + n]))
    print("MEASURED deconv fold vs two-step: logits %.3e dice %.3e worst grad l2 %.3e" % (e_log, abs(da - db_), worst))
    # bars = 2 x measured on MI355X (round 3): logits 7.2e-3, Dice 5.4e-7, worst gradient tensor 1.34e-2
    assert e_log <= 1.5e-2 and abs(da - db_) <= 1.1e-6 and worst <= 2.7e-2, (e_log, abs(da - db_), worst)
