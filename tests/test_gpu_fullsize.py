"""BASELINE config-2 FULL size (depth 4 / 32 filters, 64x128x128) checks that do not need the slow CPU oracle:
size-independent properties (linearity of the conv kernels, agreement of the bf16 MFMA path with the fp32 generic HIP path,
Dice-sum consistency, determinism of forward) on the real shapes of the benchmark."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
    assert _rel(e16.logits, e32.logits) <= 3e-2
    d16, d32 = e16.metrics_from_sums(s16), e32.metrics_from_sums(s32)
    assert abs(d16["dice_coefficient"] - d32["dice_coefficient"]) <= 2e-3
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
    assert _rel(ys.float(), ya.float() + yb.float()) <= 1e-2
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
