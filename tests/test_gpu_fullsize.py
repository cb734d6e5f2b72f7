"""BASELINE config-2 FULL size (depth 4 / 32 filters, 64x128x128) checks that do not need the slow CPU oracle:
size-independent properties (linearity of the conv kernels, agreement of the bf16 MFMA path with the fp32 generic HIP path,
Dice-sum consistency, determinism of forward) on the real shapes of the benchmark."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import bar          # noqa: E402


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_full_size_bf16_vs_fp32_paths_and_dice():
    from fmri_hip.engine import UNetEngine, UNetPlan
    spatial = (64, 128, 128)
    plan = UNetPlan(1, spatial, depth=4, n_base_filters=32)
    e16 = UNetEngine(plan, 1, dtype=torch.bfloat16, seed=42, training=False)
    e32 = UNetEngine(plan, 1, dtype=torch.float32, seed=42, training=False)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1,) + spatial + (1,), generator=g).cuda()
    y = (torch.rand((int(np.prod(spatial)),), generator=g) > 0.7).to(torch.uint8).cuda()
    e16.forward(x.to(torch.bfloat16).contiguous())
    e32.forward(x.contiguous())
    s16 = e16.loss_forward(y).cpu().numpy().copy()
    s32 = e32.loss_forward(y).cpu().numpy().copy()
    torch.cuda.synchronize()
    bar("fullsize_fwd_n1.logits_rel_bf16_vs_f32", _rel(e16.logits, e32.logits), 1.25e-2)      # measured 6.1e-3
    d16, d32 = e16.metrics_from_sums(s16), e32.metrics_from_sums(s32)
    bar("fullsize_fwd_n1.dice_abs", abs(d16["dice_coefficient"] - d32["dice_coefficient"]), 5e-6)      # measured 2.3e-6
    assert s16[7] == s32[7] == np.prod(spatial)                      # every voxel counted once
    assert s16[1] == s32[1] == float(y.sum().item())                 # sum(y) is exact in both
    # forward is deterministic (no atomics on the forward path)
    l1 = e16.logits.clone()
    e16.forward(x.to(torch.bfloat16).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(l1, e16.logits)


def test_full_size_conv_linearity_and_wgrad_paths():
    from fmri_hip import ops
    from fmri_hip._lib import IMPL_GENERIC, IMPL_MFMA
    N, D, H, W, C0, C1, Cout = 1, 64, 128, 128, 128, 64, 64          # the dec0a layer: up-sampled 128 + skip 64 -> 64
    g = torch.Generator().manual_seed(1)
    lo = lambda: (torch.randint(-4, 5, (N, D // 2, H // 2, W // 2, C0), generator=g).float() / 4).to(torch.bfloat16).cuda()
    sk = lambda: (torch.randint(-4, 5, (N, D, H, W, C1), generator=g).float() / 4).to(torch.bfloat16).cuda()
    a0, a1, b0, b1 = lo(), sk(), lo(), sk()
    w = (torch.randint(-2, 3, (27, Cout, C0 + C1), generator=g).float() / 8).to(torch.bfloat16).cuda()
    ya, yb, ys = (torch.empty((N, D, H, W, Cout), dtype=torch.bfloat16, device="cuda") for _ in range(3))
    ops.conv3d_fwd(a0, a1, w, None, ya, up0=True, act=0, impl=IMPL_MFMA)
    ops.conv3d_fwd(b0, b1, w, None, yb, up0=True, act=0, impl=IMPL_MFMA)
    ops.conv3d_fwd((a0.float() + b0.float()).to(torch.bfloat16), (a1.float() + b1.float()).to(torch.bfloat16), w, None, ys, up0=True,
                   act=0, impl=IMPL_MFMA)
    torch.cuda.synchronize()
    # small dyadic inputs: every product and partial sum is exact in fp32, sums stay below the bf16 exact-integer range / 32
    # (the sums are exact in fp32; what remains is bf16(a + b) against bf16(a) + bf16(b): one rounding step of the larger magnitudes)
    bar("fullsize_linearity.rel", _rel(ys.float(), ya.float() + yb.float()), 7.8e-3)      # measured 3.8e-3 (= 2^-8: one bf16 rounding step)
    # weight gradient: MFMA path == generic fp32-FMA path on the same bf16 data (a 16-plane slab keeps the generic kernel short)
    Dsl = 16
    dy = (torch.randint(-4, 5, (N, Dsl, H, W, Cout), generator=g).float() / 4).to(torch.bfloat16).cuda()
    a0s, a1s = a0[:, :Dsl // 2].contiguous(), a1[:, :Dsl].contiguous()
    dw_m = torch.zeros((27, Cout, C0 + C1), device="cuda")
    dw_g = torch.zeros_like(dw_m)
    db_m, db_g = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    ops.conv3d_wgrad(a0s, a1s, dy, dw_m, db_m, up0=True, impl=IMPL_MFMA)
    ops.conv3d_wgrad(a0s, a1s, dy, dw_g, db_g, up0=True, impl=IMPL_GENERIC)
    torch.cuda.synchronize()
    assert _rel(dw_m, dw_g) <= 1e-4 and _rel(db_m, db_g) <= 1e-4


def test_full_size_parity_form_equals_27_tap_kernels():
    """dec0a at BASELINE size: the parity form (8 pre-summed 2x2x2 filters on the low-res tensor) against the 27-tap kernels that read
    the up-sampled tensor, forward, both input gradients and the weight gradient, on small dyadic data (products and most partial sums
    exact, so the two formulations may only differ by the one extra bf16 rounding of the up-sampled channels' partial sum)."""
    from fmri_hip import ops
    N, D, H, W, C0, C1, Cout = 1, 64, 128, 128, 128, 64, 64
    assert ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, torch.bfloat16) == 3
    g = torch.Generator().manual_seed(2)
    bf = torch.bfloat16
    x_low = (torch.randint(-4, 5, (N, D // 2, H // 2, W // 2, C0), generator=g).float() / 4).to(bf).cuda()
    x_skip = (torch.randint(-4, 5, (N, D, H, W, C1), generator=g).float() / 4).to(bf).cuda()
    w = (torch.randint(-2, 3, (27, Cout, C0 + C1), generator=g).float() / 8).cuda()
    bias = (torch.randint(-4, 5, (Cout,), generator=g).float() / 4).cuda()
    wf, wd = torch.empty((27, Cout, C0 + C1), dtype=bf, device="cuda"), torch.empty((27, C0 + C1, Cout), dtype=bf, device="cuda")
    ops.pack_weights(w, wf, wd)
    up_f, up_d = torch.empty((8, 8, Cout, C0), dtype=bf, device="cuda"), torch.empty((8, 8, C0, Cout), dtype=bf, device="cuda")
    sk_f, sk_d = torch.empty((27, Cout, C1), dtype=bf, device="cuda"), torch.empty((27, C1, Cout), dtype=bf, device="cuda")
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
    y27, yp = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda"), torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x_low, x_skip, wf, bias, y27, up0=True, act=1)
    ops.conv3d_upcat_fwd(x_low, x_skip, up_f, sk_f, bias, yp, act=1)
    torch.cuda.synchronize()
    bar("fullsize_parity_vs_27tap.fwd_rel", _rel(yp.float(), y27.float()), 1e-2)
    dy = (torch.randint(-4, 5, (N, D, H, W, Cout), generator=g).float() / 4).to(bf).cuda()
    cat = torch.empty((N, D, H, W, C0 + C1), dtype=bf, device="cuda")
    dl27, dlp, dsp = torch.empty_like(x_low), torch.empty_like(x_low), torch.empty_like(x_skip)
    ops.conv3d_dgrad(dy, wd, cat)
    ops.upsample_bwd(cat, dl27, 0, xmask=x_low)
    ops.conv3d_upcat_dgrad(dy, up_d, sk_d, x_low, None, dlp, dsp)
    torch.cuda.synchronize()
    # the 27-tap path rounds the full-resolution gradient to bf16 before the 2x2x2 reduction, the parity form does not
    bar("fullsize_parity_vs_27tap.dlow_rel", _rel(dlp.float(), dl27.float()), 1.4e-2)      # measured 7.1e-3
    bar("fullsize_parity_vs_27tap.dskip_rel", _rel(dsp.float(), cat[..., C0:].float()), 0.0)      # same taps, same order: identical
    Dsl = 16
    dys, xls, xss = dy[:, :Dsl].contiguous(), x_low[:, :Dsl // 2].contiguous(), x_skip[:, :Dsl].contiguous()
    dw27, dwp = torch.zeros((27, Cout, C0 + C1), device="cuda"), torch.zeros((27, Cout, C0 + C1), device="cuda")
    db27, dbp = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    ops.conv3d_wgrad(xls, xss, dys, dw27, db27, up0=True)
    ops.conv3d_upcat_wgrad(xls, xss, dys, dwp, dbp, torch.empty(64 * Cout * C0, device="cuda"))
    torch.cuda.synchronize()
    assert _rel(dwp, dw27) <= 1e-4 and _rel(dbp, db27) <= 1e-4


def test_bench_contract_smoke():
    """bench.py runs end to end on one GPU and prints the driver's JSON contract (+ roofline with live HIP-event timing)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                                  stderr=subprocess.STDOUT, timeout=600).decode()
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "bf16" and j["value"] > 10
    r = j["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the parity-mode leg runs the fp32 step on the MFMA kernels; without the cpu_baseline leg it has no oracle reference to compare with
    pm = j["parity_mode"]
    assert pm["dtype"] == "fp32" and pm["patches_per_s"] > 1 and pm["logits_rel"] is None and pm["kernels"].startswith("fp32 instantiation")


WS_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join(%r, "fetal-mri-segmentation_amd"))
import numpy as np, torch
from fmri_hip import ops
out = {}
g = torch.Generator().manual_seed(1)
def rnd(shape, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).cuda().to(torch.bfloat16)
cases = [  # N, D, H, W, C0, C1, Cout, up0, mask
    (1, 4, 8, 16, 32, 0, 32, 0, 0), (2, 8, 16, 32, 64, 0, 64, 0, 1), (1, 8, 16, 16, 96, 32, 96, 0, 0), (1, 8, 16, 32, 128, 64, 64, 1, 0),
    (3, 4, 24, 48, 64, 64, 128, 0, 1), (1, 12, 8, 16, 256, 0, 160, 0, 0), (5, 4, 8, 16, 32, 0, 64, 0, 0),
]
for i, (N, D, H, W, C0, C1, Cout, up0, msk) in enumerate(cases):
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    x0, x1 = rnd(s0), (rnd((N, D, H, W, C1)) if C1 else None)
    w = rnd((27, Cout, C0 + C1), 0.05)
    b = torch.randn(Cout, generator=g).cuda()
    y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
    ops.conv3d_fwd(x0, x1, w, b, y, up0=bool(up0), act=1)
    out["fwd%%d" %% i] = y.view(torch.int16).cpu().numpy()
    if not up0 and not C1:
        wd = rnd((27, C0, Cout), 0.05)
        dy = rnd((N, D, H, W, Cout))
        dx = torch.empty((N, D, H, W, C0), device="cuda", dtype=torch.bfloat16)
        ops.conv3d_dgrad(dy, wd, dx, mask=(rnd((N, D, H, W, C0)) if msk else None))
        out["dgrad%%d" %% i] = dx.view(torch.int16).cpu().numpy()
# parity form (MODE 1 + residual epilogue, MODE 2)
N, D, H, W, C0, C1, Cout = 2, 8, 16, 32, 128, 64, 64
xl, xs = rnd((N, D // 2, H // 2, W // 2, C0)), rnd((N, D, H, W, C1))
w = rnd((27, Cout, C0 + C1), 0.05).float().contiguous()
up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=torch.bfloat16); up_d = torch.empty((8, 8, C0, Cout), device="cuda", dtype=torch.bfloat16)
sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=torch.bfloat16); sk_d = torch.empty((27, C1, Cout), device="cuda", dtype=torch.bfloat16)
ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
ops.conv3d_upcat_fwd(xl, xs, up_f, sk_f, torch.randn(Cout, generator=g).cuda(), y, act=1)
dy = rnd((N, D, H, W, Cout)); dxl, dxs = torch.empty_like(xl), torch.empty_like(xs)
ops.conv3d_upcat_dgrad(dy, up_d, sk_d, rnd(tuple(xl.shape)), None, dxl, dxs)
torch.cuda.synchronize()
out["up_fwd"], out["up_dxl"], out["up_dxs"] = (t.view(torch.int16).cpu().numpy() for t in (y, dxl, dxs))
# the same with more tiles than CUs (288): the skip launch takes the asynchronous residual epilogue unless FMRI_RES_ASYNC=0
N, D, H, W = 3, 16, 32, 96
xl, xs = rnd((N, D // 2, H // 2, W // 2, C0)), rnd((N, D, H, W, C1))
y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
ops.conv3d_upcat_fwd(xl, xs, up_f, sk_f, torch.randn(Cout, generator=g).cuda(), y, act=2, alpha=0.01)
torch.cuda.synchronize()
out["up_fwd_big"] = y.view(torch.int16).cpu().numpy()
np.savez(sys.argv[1], **out)
print("DONE")
"""


def test_warp_specialised_kernel_is_bit_identical_to_the_symmetric_one(tmp_path):
    """k_conv_fwd_ws (4 MFMA waves + 4 LDS-DMA waves) against k_conv_fwd_mfma (FMRI_FWD_WS=0) on the same seeded inputs: both accumulate
    every output in the same (chunk, (kd,kh), kw, k-step) order, so forward, input gradients (with and without mask), dual-source /
    fused-upsample reads and the parity-form launches must agree BIT FOR BIT - any race in the producer/consumer hand-over shows up here.
    Third run: FMRI_RES_ASYNC=0 (the skip launch's residual added by the MFMA waves instead of the LDS-DMA waves' drain) - the same bits
    again, so a result does not depend on which form the grid size selects."""
    import subprocess, sys, os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = tmp_path / "ws.py"
    f.write_text(WS_SCRIPT % root)
    outs = []
    for ws, ra in (("1", "1"), ("0", "1"), ("1", "0")):
        o = str(tmp_path / ("out%s%s.npz" % (ws, ra)))
        r = subprocess.run([sys.executable, str(f), o], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, FMRI_FWD_WS=ws, FMRI_RES_ASYNC=ra))
        assert r.returncode == 0 and "DONE" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
        outs.append(np.load(o))
    assert len(outs[0].files) >= 14
    for other in outs[1:]:
        assert set(outs[0].files) == set(other.files)
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], other[k]), k


def test_bench_contract_line(tmp_path):
    """`python bench.py --steps 3 --warmup 1` prints ONE JSON line with the driver's contract keys, the two roofline objects of the
    two-stream step and a cpu-free run when asked (the CPU baseline itself takes ~25 s and is exercised by the default bench run)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_exclusive", "step_mfma_frac", "kernel_ms_per_step"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "patches/s" and d["dtype"] == "bf16" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    for ro in (d["roofline"], d["roofline_exclusive"]):
        assert ro["bound"] == "mfma" and ro["peak"] == 2500.0 and ro["unit"] == "TFLOP/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
        assert ro["launches_per_step"] == 32 and ro["avg_launch_ms"] > 0
    assert d["roofline_exclusive"]["frac"] > d["roofline"]["frac"]          # exclusive kernel time < the bracket inside the two-stream region
    assert 0.3 < d["step_mfma_frac"] < 1.0
