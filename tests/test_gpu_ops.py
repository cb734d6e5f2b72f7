"""GPU parity of each C-ABI kernel against torch-CPU fp64 restatements (checker).  Run with -m gpu on an MI355X.

Tolerances (each <= 2-5x the error measured on MI355X with FMRI_MEASURE=1, profiles/r02_tolerance_use.json): |got - ref| <=
atol * max|ref| + rtol * |ref|.  fp32 mode: plain fmaf chains, only the summation order differs (measured ~3e-7) -> 2e-6.  bf16 mode:
the inputs are bf16 and the checker uses them in fp64, so what remains is ONE rounding of the stored output - bf16 keeps 8 significant
bits, i.e. up to 2^-8 = 3.9e-3 relative - plus the fp32 accumulation order (~2e-5 of the largest output): rtol 5e-3, atol 1e-4.  One
dropped (tap, channel) product at Cin = 32 is ~3 % of an output's sigma: 100x outside these bars.  Weight gradients are fp32 sums of
exact bf16 products (measured 4e-7) -> 2e-6.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import (assert_close, f64, keras_kernel_from_packed, planar_kernel, ref_concat_input, ref_conv_fwd, rnd, to_ncdhw,
                      to_ndhwc)

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-6, 2e-6), torch.bfloat16: (5e-3, 1e-4)}
WG_TOL = {torch.float32: (1e-5, 2e-6), torch.bfloat16: (2e-6, 2e-6)}      # weight / bias gradients (fp32 outputs)
PARITY_TOL = (6e-3, 6e-3)      # parity form: the up-sampled channels' partial sum is stored as bf16 once more (2^-8 of a partial that may exceed the result)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from fmri_hip import ops as o
    return o


CONV_CASES = [
    # name, N,D,H,W, C0, up0, C1, Cout, impl
    ("generic_small_f32", torch.float32, 1, 4, 8, 16, 3, False, 0, 5, 1),
    ("generic_odd_f32", torch.float32, 2, 3, 5, 7, 2, False, 0, 3, 1),
    ("generic_dual_up_f32", torch.float32, 1, 4, 8, 16, 8, True, 4, 6, 1),
    ("generic_cin1_f32", torch.float32, 1, 4, 16, 16, 1, False, 0, 8, 1),
    ("generic_2d_f32", torch.float32, 2, 1, 16, 16, 5, False, 0, 8, 1),
    ("generic_bf16", torch.bfloat16, 1, 4, 8, 16, 8, False, 0, 16, 1),
    ("mfma_32_64", torch.bfloat16, 1, 4, 8, 16, 32, False, 0, 64, 2),
    ("mfma_64_32", torch.bfloat16, 1, 4, 8, 16, 64, False, 0, 32, 2),
    ("mfma_multi_tile", torch.bfloat16, 2, 8, 16, 32, 64, False, 0, 64, 2),
    ("mfma_dual_up", torch.bfloat16, 1, 8, 16, 32, 64, True, 32, 64, 2),
    ("mfma_dual_noup_128", torch.bfloat16, 1, 4, 8, 16, 32, False, 64, 128, 2),
    # 8 channel chunks on an all-boundary tile: the out-of-volume halo rows keep reading zeros as the source pointers advance
    ("mfma_deep_256", torch.bfloat16, 1, 4, 8, 16, 256, False, 0, 64, 2),
    ("mfma_dual_up_deep", torch.bfloat16, 1, 8, 16, 16, 128, True, 64, 96, 2),
    # 8x8x8 workgroup tiles: volumes whose W is a multiple of 8 but not of 16 (deepest level of deep models)
    ("mfma_cube_8", torch.bfloat16, 2, 8, 8, 8, 64, False, 0, 64, 2),
    ("mfma_cube_multi", torch.bfloat16, 1, 16, 8, 24, 96, False, 0, 32, 2),
    ("mfma_cube_dual_up", torch.bfloat16, 1, 8, 16, 8, 32, True, 32, 64, 2),
    ("first_layer_4_modalities", torch.bfloat16, 1, 4, 16, 32, 4, False, 0, 32, 0),
    ("first_layer_2_modalities", torch.bfloat16, 1, 8, 16, 32, 2, False, 0, 64, 0),
    ("first_layer_32", torch.bfloat16, 2, 4, 16, 32, 1, False, 0, 32, 0),
    ("first_layer_64_multi", torch.bfloat16, 1, 8, 32, 64, 1, False, 0, 64, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_fwd(ops, case):
    name, dtype, N, D, H, W, C0, up0, C1, Cout, impl = case
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = rnd(s0, 1, dtype)
    src1 = rnd((N, D, H, W, C1), 2, dtype) if C1 else None
    w = rnd((27, Cout, C0 + C1), 3, dtype, scale=0.2)
    bias = rnd((Cout,), 4, torch.float32)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_fwd(src0, src1, w, bias, y, up0=up0, act=1, impl=impl)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(src0), None if src1 is None else f64(src1), up0, f64(w), f64(bias), 1)
    assert_close(y, ref, *TOL[dtype], what=name)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_fwd_noact_mask(ops, case):
    """the dgrad form: no bias, no activation, ReLU mask epilogue"""
    name, dtype, N, D, H, W, C0, up0, C1, Cout, impl = case
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = rnd(s0, 5, dtype)
    src1 = rnd((N, D, H, W, C1), 6, dtype) if C1 else None
    w = rnd((27, Cout, C0 + C1), 7, dtype, scale=0.2)
    mask = rnd((N, D, H, W, Cout), 8, dtype)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_fwd(src0, src1, w, None, y, up0=up0, act=0, mask=mask, impl=impl)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(src0), None if src1 is None else f64(src1), up0, f64(w), None, 0)
    ref = ref * (f64(mask) > 0)
    assert_close(y, ref, *TOL[dtype], what=name)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_wgrad(ops, case):
    name, dtype, N, D, H, W, C0, up0, C1, Cout, impl = case
    if impl == 2 and (Cout % 64 or W % 16):
        pytest.skip("MFMA wgrad needs Cout % 64 == 0 and W % 16 == 0")
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = rnd(s0, 9, dtype)
    src1 = rnd((N, D, H, W, C1), 10, dtype) if C1 else None
    dy = rnd((N, D, H, W, Cout), 11, dtype)
    dw = torch.zeros((27, Cout, C0 + C1), dtype=torch.float32, device="cuda")
    db = torch.zeros((Cout,), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(src0, src1, dy, dw, db, up0=up0, impl=impl)
    torch.cuda.synchronize()
    nws = ops.conv3d_wgrad_workspace_bytes(C0, C1, Cout, N, D, H, W, dtype)
    if impl == 2:
        # slab flush (plain stores + reduction pass) gives the same sums as the atomic flush, and is bit-reproducible
        assert nws > 0
        ws = torch.empty(nws // 4, dtype=torch.float32, device="cuda")
        outs = []
        for _ in range(2):
            dw2 = torch.zeros_like(dw)
            db2 = torch.zeros_like(db)
            ops.conv3d_wgrad(src0, src1, dy, dw2, db2, up0=up0, impl=impl, workspace=ws)
            outs.append(dw2)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
        assert_close(outs[0], dw, 1e-6, 1e-6, what=name + " slab vs atomic")
    x = ref_concat_input(f64(src0), None if src1 is None else f64(src1), up0)
    wk = torch.zeros((Cout, C0 + C1, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    out = F.conv3d(x, wk, None, padding=1)
    out.backward(to_ncdhw(f64(dy)))
    ref = wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
    rt = WG_TOL[dtype]
    assert_close(dw, ref, *rt, what=name + " dw")
    assert_close(db, f64(dy).sum(dim=(0, 1, 2, 3)), *rt, what=name + " db")


@pytest.mark.parametrize("dtype,Cin,Cout,impl", [(torch.float32, 5, 7, 1), (torch.bfloat16, 32, 64, 2), (torch.bfloat16, 64, 32, 2),
                                                 (torch.bfloat16, 192, 64, 2)])
def test_pack_and_dgrad(ops, dtype, Cin, Cout, impl):
    """dgrad through the tap-flipped transposed copy == autograd of the forward conv wrt its input"""
    N, D, H, W = 1, 4, 8, 16
    wm = rnd((27, Cout, Cin), 12, torch.float32, scale=0.2)
    wf = torch.empty((27, Cout, Cin), dtype=dtype, device="cuda")
    wd = torch.empty((27, Cin, Cout), dtype=dtype, device="cuda")
    ops.pack_weights(wm, wf, wd)
    dy = rnd((N, D, H, W, Cout), 13, dtype)
    mask = rnd((N, D, H, W, Cin), 14, dtype)
    dx = torch.full((N, D, H, W, Cin), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_dgrad(dy, wd, dx, mask=mask, impl=impl)
    torch.cuda.synchronize()
    assert torch.equal(wf.float().cpu(), wm.to(dtype).float().cpu())
    x = torch.zeros((N, Cin, D, H, W), dtype=torch.float64, requires_grad=True)
    out = F.conv3d(x, keras_kernel_from_packed(f64(wf)), None, padding=1)
    out.backward(to_ncdhw(f64(dy)))
    ref = to_ndhwc(x.grad) * (f64(mask) > 0)
    assert_close(dx, ref, *TOL[dtype], what="dgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [1, 6, 16, 64])
def test_maxpool_fwd_bwd(ops, dtype, C):
    N, D, H, W = 2, 4, 6, 8
    x = torch.relu(rnd((N, D, H, W, C), 20, dtype))  # post-ReLU like in the network (ties at 0)
    y = torch.empty((N, D // 2, H // 2, W // 2, C), dtype=dtype, device="cuda")
    ops.maxpool_fwd(x, y)
    xr = to_ncdhw(f64(x)).requires_grad_(True)
    yr = F.max_pool3d(xr, 2)
    torch.cuda.synchronize()
    assert torch.equal(f64(y), to_ndhwc(yr.detach()))
    dy = rnd(tuple(y.shape), 21, dtype)
    add = rnd((N, D, H, W, C + 8), 22, dtype)
    dx = torch.full_like(x, float("nan"))
    ops.maxpool_bwd(x, dy, dx, add=add, add_off=8, relu_mask=True)
    torch.cuda.synchronize()
    yr.backward(to_ncdhw(f64(dy)))
    ref = (to_ndhwc(xr.grad) + f64(add)[..., 8:]) * (f64(x) > 0)
    assert_close(dx, ref, 1e-6 if dtype == torch.float32 else 5e-3, 1e-6 if dtype == torch.float32 else 1e-3, what="maxpool_bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_fwd_bwd(ops, dtype):
    N, D, H, W, C = 2, 2, 3, 4, 8
    x = rnd((N, D, H, W, C), 30, dtype)
    y = torch.zeros((N, 2 * D, 2 * H, 2 * W, C + 4), dtype=dtype, device="cuda")
    ops.upsample_fwd(x, y, y_off=4)
    torch.cuda.synchronize()
    ref = to_ndhwc(ref_concat_input(f64(x), None, True))
    assert torch.equal(f64(y)[..., 4:], ref) and float(f64(y)[..., :4].abs().max()) == 0
    dy = rnd(tuple(y.shape), 31, dtype)
    dx = torch.empty_like(x)
    ops.upsample_bwd(dy, dx, dy_off=4, xmask=x)
    torch.cuda.synchronize()
    g = f64(dy)[..., 4:].reshape(N, D, 2, H, 2, W, 2, C).sum(dim=(2, 4, 6)) * (f64(x) > 0)
    assert_close(dx, g, *TOL[dtype], what="upsample_bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,L", [(64, 1), (16, 2), (6, 1)])
def test_conv1x1_fwd_bwd(ops, dtype, C, L):
    nv = 1000
    x = torch.relu(rnd((nv, C), 40, dtype))
    w = rnd((L, C), 41, torch.float32)
    b = rnd((L,), 42, torch.float32)
    logits = torch.empty((nv, L), dtype=torch.float32, device="cuda")
    ops.conv1x1_fwd(x, w, b, logits)
    torch.cuda.synchronize()
    ref = f64(x) @ f64(w).T + f64(b)
    assert_close(logits, ref, 1e-5, 1e-5, what="conv1x1_fwd")
    dl = rnd((nv, L), 43, torch.float32)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(w)
    db = torch.zeros_like(b)
    ops.conv1x1_bwd(x, w, dl, dx, dw, db, relu_mask=True)
    torch.cuda.synchronize()
    assert_close(dx, (f64(dl) @ f64(w)) * (f64(x) > 0), *TOL[dtype], what="conv1x1 dx")
    assert_close(dw, f64(dl).T @ f64(x), 1e-4, 1e-5, what="conv1x1 dw")
    assert_close(db, f64(dl).sum(0), 1e-4, 1e-5, what="conv1x1 db")


def test_sigmoid_dice_fwd_bwd(ops):
    from oracle import metrics_oracle as M
    n = 5000
    logits = rnd((n,), 50, torch.float32, scale=2.0)
    y = (torch.rand(n, generator=torch.Generator().manual_seed(51)) > 0.7).to(torch.uint8).cuda()
    probs = torch.empty_like(logits)
    sums = torch.zeros(16, dtype=torch.float64, device="cuda")
    ops.sigmoid_dice_fwd(logits, y, probs, sums)
    dl = torch.empty_like(logits)
    ops.sigmoid_dice_bwd(probs, y, sums, dl)
    torch.cuda.synchronize()
    p64 = torch.sigmoid(f64(logits)).requires_grad_(False)
    s = sums.cpu().numpy()
    dice = (2 * s[0] + 1) / (s[1] + s[2] + 1)
    assert dice == pytest.approx(M.dice_coefficient(y.cpu().numpy(), p64.numpy()), rel=1e-6)
    vod = (s[3] + 1) / (s[4] + s[5] - s[3] + 1)
    assert vod == pytest.approx(M.vod_coefficient(y.cpu().numpy(), p64.numpy()), rel=1e-6)
    assert s[6] / s[7] == pytest.approx(M.binary_accuracy(y.cpu().numpy(), p64.numpy()), abs=2e-4)
    z = f64(logits).requires_grad_(True)
    yy = f64(y)
    pp = torch.sigmoid(z)
    loss = -(2 * (yy * pp).sum() + 1) / (yy.sum() + pp.sum() + 1)
    loss.backward()
    assert_close(dl, z.grad, 1e-4, 1e-5, what="dice bwd")


def test_adam(ops):
    from oracle.unet_oracle import KerasAdam
    n = 1003
    p = rnd((1024,), 60, torch.float32)[:n].contiguous()
    W = {"a": p.cpu().numpy().astype(np.float64)}
    opt = KerasAdam(W, lr=1e-3)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    import math
    for t in range(1, 4):
        g = rnd((n,), 60 + t, torch.float32)
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_step(p, g, m, v, lr_t)
        opt.step(W, {"a": g.cpu().numpy().astype(np.float64)})
    torch.cuda.synchronize()
    assert_close(p, torch.tensor(W["a"]), 1e-5, 1e-6, what="adam")


def test_tiles(ops):
    rs = np.random.RandomState(70)
    vol = torch.tensor(rs.randn(20, 18, 12).astype(np.float32)).cuda()
    idx = torch.tensor([[0, 0, 0], [12, 10, 4], [5, 3, 2]], dtype=torch.int32).cuda()
    patch = (8, 8, 8)
    tiles = torch.empty((3,) + patch, dtype=torch.float32, device="cuda")
    ops.tile_gather(vol, idx, patch, tiles)
    torch.cuda.synchronize()
    for b, (x, y, z) in enumerate(idx.cpu().tolist()):
        assert torch.equal(tiles[b].cpu(), vol[x:x + 8, y:y + 8, z:z + 8].cpu())
    acc = torch.zeros((20, 18, 12, 2), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((20, 18, 12), dtype=torch.int32, device="cuda")
    pred = torch.tensor(rs.randn(3, 8, 8, 8, 2).astype(np.float32)).cuda()
    ops.tile_scatter_accumulate(pred, idx, patch, acc, cnt)
    out = torch.empty_like(acc)
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.tile_finalize(acc, cnt, out, bad)
    torch.cuda.synchronize()
    ra = np.zeros((20, 18, 12, 2))
    rc = np.zeros((20, 18, 12))
    for b, (x, y, z) in enumerate(idx.cpu().tolist()):
        ra[x:x + 8, y:y + 8, z:z + 8] += pred[b].cpu().numpy().astype(np.float64)
        rc[x:x + 8, y:y + 8, z:z + 8] += 1
    np.testing.assert_allclose(acc.cpu().numpy(), ra, atol=1e-12)
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert int(bad.item()) == int((rc == 0).sum())


# ------------------------------------------------------------------------------------------------ planar (2-D slice) semantics
PLANAR_CASES = [
    # name, dtype, slices, H, W, C0, up0, C1, Cout, impl
    ("planar_generic_f32", torch.float32, 3, 8, 16, 5, False, 0, 6, 1),
    ("planar_generic_dual_up_f32", torch.float32, 2, 8, 16, 4, True, 3, 5, 1),
    ("planar_mfma", torch.bfloat16, 8, 16, 32, 64, False, 0, 64, 2),
    ("planar_mfma_dual_up", torch.bfloat16, 4, 16, 32, 64, True, 32, 64, 2),
    ("planar_mfma_32", torch.bfloat16, 4, 8, 16, 32, False, 0, 32, 2),
    # first layer of the 2-D models: a stack of 5 (3, 7, 1) slices as channels; the (tap, channel) pairs are the MFMA k-dimension
    ("planar_first_5", torch.bfloat16, 8, 32, 64, 5, False, 0, 32, 0),
    ("planar_first_3", torch.bfloat16, 4, 16, 32, 3, False, 0, 64, 0),
    ("planar_first_7", torch.bfloat16, 4, 16, 64, 7, False, 0, 32, 0),
    ("planar_first_1", torch.bfloat16, 4, 32, 32, 1, False, 0, 32, 0),
]


@pytest.mark.parametrize("case", PLANAR_CASES, ids=[c[0] for c in PLANAR_CASES])
def test_planar_conv_fwd_dgrad_wgrad(ops, case):
    name, dtype, S, H, W, C0, up0, C1, Cout, impl = case
    s0 = (1, S, H // 2, W // 2, C0) if up0 else (1, S, H, W, C0)
    src0 = rnd(s0, 1, dtype)
    src1 = rnd((1, S, H, W, C1), 2, dtype) if C1 else None
    w = rnd((27, Cout, C0 + C1), 3, dtype, scale=0.2)
    bias = rnd((Cout,), 4, torch.float32)
    y = torch.full((1, S, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_fwd(src0, src1, w, bias, y, up0=up0, act=1, impl=impl, planar=True)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(src0), None if src1 is None else f64(src1), up0, f64(w), f64(bias), 1, planar=True)
    assert_close(y, ref, *TOL[dtype], what=name + " fwd")
    # weight gradient: only the centre kd plane exists
    if not (impl == 2 and Cout % 64):
        dy = rnd((1, S, H, W, Cout), 11, dtype)
        dw = torch.zeros((27, Cout, C0 + C1), dtype=torch.float32, device="cuda")
        db = torch.zeros((Cout,), dtype=torch.float32, device="cuda")
        ops.conv3d_wgrad(src0, src1, dy, dw, db, up0=up0, impl=impl, planar=True)
        torch.cuda.synchronize()
        x = ref_concat_input(f64(src0), None if src1 is None else f64(src1), up0, planar=True)
        wk = torch.zeros((Cout, C0 + C1, 3, 3, 3), dtype=torch.float64, requires_grad=True)
        F.conv3d(x, wk, None, padding=1).backward(to_ncdhw(f64(dy)))
        refg = planar_kernel(wk.grad).permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
        rt = WG_TOL[dtype]
        assert_close(dw, refg, *rt, what=name + " dw")
        assert_close(db, f64(dy).sum(dim=(0, 1, 2, 3)), *rt, what=name + " db")
    # input gradient through the packed transposed copy
    if not up0 and not C1:
        wm = w.float()
        wd = torch.empty((27, C0, Cout), dtype=dtype, device="cuda")
        ops.pack_weights(wm.contiguous(), None, wd)
        dy = rnd((1, S, H, W, Cout), 12, dtype)
        dx = torch.full((1, S, H, W, C0), float("nan"), dtype=dtype, device="cuda")
        ops.conv3d_dgrad(dy, wd, dx, impl=impl, planar=True)
        torch.cuda.synchronize()
        xr = torch.zeros((1, C0, S, H, W), dtype=torch.float64, requires_grad=True)
        F.conv3d(xr, planar_kernel(keras_kernel_from_packed(f64(w))), None, padding=1).backward(to_ncdhw(f64(dy)))
        assert_close(dx, to_ndhwc(xr.grad), *TOL[dtype], what=name + " dgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_planar_pool_and_upsample(ops, dtype):
    S, H, W, C = 3, 6, 8, 8
    x = torch.relu(rnd((1, S, H, W, C), 20, dtype))
    y = torch.empty((1, S, H // 2, W // 2, C), dtype=dtype, device="cuda")
    ops.maxpool_fwd(x, y, planar=True)
    xr = f64(x)[0].permute(0, 3, 1, 2).contiguous().requires_grad_(True)            # (S,C,H,W)
    yr = F.max_pool2d(xr, 2)
    torch.cuda.synchronize()
    assert torch.equal(f64(y)[0], yr.detach().permute(0, 2, 3, 1))
    dy = rnd(tuple(y.shape), 21, dtype)
    dx = torch.full_like(x, float("nan"))
    ops.maxpool_bwd(x, dy, dx, relu_mask=True, planar=True)
    torch.cuda.synchronize()
    yr.backward(f64(dy)[0].permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1) * (f64(x)[0] > 0)
    assert_close(dx[0], ref, 1e-6 if dtype == torch.float32 else 5e-3, 1e-6 if dtype == torch.float32 else 1e-3, what="planar maxpool bwd")
    g = rnd((1, S, H, W, C), 22, dtype)
    lo = torch.empty((1, S, H // 2, W // 2, C), dtype=dtype, device="cuda")
    ops.upsample_bwd(g, lo, planar=True)
    up = torch.zeros((1, S, H, W, C), dtype=dtype, device="cuda")
    ops.upsample_fwd(y, up, planar=True)
    torch.cuda.synchronize()
    assert_close(lo, f64(g).reshape(1, S, H // 2, 2, W // 2, 2, C).sum(dim=(3, 5)), *TOL[dtype], what="planar upsample bwd")
    assert torch.equal(f64(up), f64(y).repeat_interleave(2, dim=2).repeat_interleave(2, dim=3))


# ------------------------------------------------------------------------------------------------ normalisation + deconvolution
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["batch", "instance"])
@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("C", [16, 512])          # 512: the widest layer of BASELINE configs[1] (64 channel groups x 4 voxel lanes per workgroup)
def test_norm_act_fwd_bwd(ops, dtype, mode, act, C):
    from oracle import unet_oracle as O
    N, D, H, W = 2, 4, 6, 8
    alpha = 0.3
    x = rnd((N, D, H, W, C), 80, dtype, scale=1.5) + 0.3
    gamma = rnd((C,), 81, torch.float32) * 0.5 + 1.0
    beta = rnd((C,), 82, torch.float32) * 0.2
    per = mode == "instance"
    G = N if per else 1
    stats = torch.zeros((G, C, 3), device="cuda")
    ws = torch.zeros((G, C, 2), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    ops.norm_act_fwd(x, gamma, beta, y, stats, ws, per, eps=1e-3, eps_on_std=per, act=act, alpha=alpha)
    xr = to_ncdhw(f64(x)).requires_grad_(True)
    gr, br = f64(gamma).requires_grad_(True), f64(beta).requires_grad_(True)
    z = O._instancenorm(xr, gr, br) if per else O._batchnorm_train(xr, gr, br)
    yr = F.relu(z) if act == 1 else F.leaky_relu(z, alpha)
    torch.cuda.synchronize()
    tol = (1e-4, 1e-5) if dtype == torch.float32 else (5e-3, 2e-4)
    assert_close(y, to_ndhwc(yr.detach()), *tol, what="norm fwd")
    dy = rnd((N, D, H, W, C), 83, dtype)
    # the backward reads the STORED y (bf16-rounded in bf16 mode) for act': mirror that in the checker
    dx = torch.empty_like(x)
    dg = torch.zeros(C, device="cuda")
    db = torch.zeros(C, device="cuda")
    ops.norm_act_bwd(x, y, dy, gamma, stats, dx, dg, db, ws, per, act=act, alpha=alpha)
    torch.cuda.synchronize()
    yr.backward(to_ncdhw(f64(dy)))
    tolb = (2e-4, 2e-5) if dtype == torch.float32 else (6e-3, 2e-3)
    assert_close(dx, to_ndhwc(xr.grad), *tolb, what="norm dx")
    # the form that recomputes the sign of the output from x instead of reading y: the same operations as the forward, the same bits
    dx2, dg2, db2 = torch.empty_like(x), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.norm_act_bwd(x, None, dy, gamma, stats, dx2, dg2, db2, ws, per, act=act, alpha=alpha, beta=beta)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx), "norm backward from x differs from the one reading y: %g" % float((dx2.float() - dx.float()).abs().max())
    assert_close(dg2, dg, 1e-6, 1e-6, what="norm dgamma (from x)")
    assert_close(dg, gr.grad, *tolb, what="norm dgamma")
    assert_close(db, br.grad, *tolb, what="norm dbeta")


@pytest.mark.parametrize("shape", [(1, 16, 32, 64, 32, 64), (2, 8, 16, 32, 64, 32)], ids=lambda c: "N%d_%dx%dx%d_%d_%d" % c)
def test_stride2_conv_on_the_parity_kernels_is_exact_on_dyadic_data(ops, shape):
    """fmri_hip.strided_parity: Conv3D(3x3x3, strides 2, 'same') = the up-backward gather launch over the input with the filter's 27 taps in
    27 of its 64 (parity, block offset) slots; its input gradient = the up-forward scatter launch; its weight gradient = 27 of the 64 slot
    gradients.  Small dyadic values: every sum is exact in fp32, so all three must equal torch's strided convolution bit for bit (TF 'same'
    on even dims pads one plane BEHIND the volume: out[o] = sum_t W[t] x[2o + t])."""
    from fmri_hip.strided_parity import StridedParity
    N, D, H, W, Cin, Cout = shape
    bf = torch.bfloat16
    ok = ops.conv3d_upcat_ok(Cout, 0, Cin, D, H, W, bf)          # bit 1 (weight gradient): needs 64-wide blocks of the input's channels
    assert ok & 1
    g = torch.Generator().manual_seed(sum(shape))
    dy4 = lambda sh, lo, hi, div: torch.randint(lo, hi, sh, generator=g).float() / div
    x = dy4((N, D, H, W, Cin), -4, 5, 4.0)
    w = dy4((27, Cout, Cin), -2, 3, 8.0)
    sp = StridedParity("cuda")
    fwd_img = torch.zeros((8, 8, Cout, Cin), device="cuda", dtype=bf)
    dg_img = torch.zeros((8, 8, Cin, Cout), device="cuda", dtype=bf)
    sp.pack(w.cuda(), fwd_img, dg_img)
    y = torch.empty((N, D // 2, H // 2, W // 2, Cout), device="cuda", dtype=bf)
    ops.conv3d_upcat_dgrad(x.to(bf).cuda(), fwd_img, None, None, None, y, None)
    xr = to_ncdhw(x).requires_grad_(True)
    wk = keras_kernel_from_packed(w).float().requires_grad_(True)
    ref = F.conv3d(F.pad(xr, (0, 1, 0, 1, 0, 1)), wk, None, stride=2)
    assert torch.equal(y.cpu().view(torch.int16), to_ndhwc(ref.detach()).to(bf).view(torch.int16)), "forward"
    # the entry point the Isensee engine calls: the same launch with the conv's fp32 bias in the accumulators (one rounding, of the sum)
    b = dy4((Cout,), -8, 9, 16.0) + 2.0 ** -12                          # not representable in bf16: a bias rounded on its own would show
    yb = torch.empty_like(y)
    ops.conv3d_stride2_fwd(x.to(bf).cuda(), fwd_img, b.cuda(), yb)
    refb = F.conv3d(F.pad(xr, (0, 1, 0, 1, 0, 1)), wk, b, stride=2)
    assert torch.equal(yb.cpu().view(torch.int16), to_ndhwc(refb.detach()).to(bf).view(torch.int16)), "forward with bias"
    dy = dy4((N, D // 2, H // 2, W // 2, Cout), -2, 3, 2.0)
    ref.backward(to_ncdhw(dy))
    dx = torch.empty((N, D, H, W, Cin), device="cuda", dtype=bf)
    ops.conv3d_upcat_fwd(dy.to(bf).cuda(), None, dg_img, None, None, dx, act=0)
    assert torch.equal(dx.cpu().view(torch.int16), to_ndhwc(xr.grad).to(bf).view(torch.int16)), "input gradient"
    if not ok & 2:
        return
    dw27 = torch.zeros((27, Cin, Cout), device="cuda")
    db = torch.zeros(Cin, device="cuda")
    dwc = torch.zeros(64 * Cin * Cout, device="cuda")
    ops.conv3d_upcat_wgrad(dy.to(bf).cuda(), None, x.to(bf).cuda(), dw27, db, dwc)
    dw = sp.unpack_wgrad(dwc, Cout, Cin)
    ref_dw = wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, Cin)
    assert torch.equal(dw.cpu(), ref_dw), "weight gradient: max diff %g" % float((dw.cpu() - ref_dw).abs().max())


# (N, D, H, W, Cin, Cout): more (tile, channel block) pairs than CUs - 64-wide blocks (Cout 128 -> 2 per tile), 32-wide (Cout 32)
NTAIL_CASES = [(2, 16, 64, 64, 32, 128), (3, 16, 64, 64, 64, 32)]


@pytest.mark.parametrize("case", NTAIL_CASES, ids=lambda c: "N%d_%dx%dx%d_%d_%d" % c)
@pytest.mark.parametrize("per", [0, 1])
def test_conv_fwd_stats_tail(ops, case, per):
    """fmri_conv3d_fwd_stats: the output is the plain launch's bit for bit, and ws holds the sums of the stored bf16 values and of their
    squares per (group, channel) - what the normalisation's own reduction pass would compute (fp64 sums of the same values as reference)"""
    N, D, H, W, Cin, Cout = case
    bf = torch.bfloat16
    assert ops.conv3d_fwd_ntail_ok(Cin, 0, Cout, N, D, H, W, bf)
    x = rnd((N, D, H, W, Cin), 300, bf)
    w = rnd((27, Cout, Cin), 301, bf, scale=0.06)
    b = rnd((Cout,), 302, torch.float32)
    y0, y1 = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda"), torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x, None, w, b, y0, act=0)
    G = N if per else 1
    wsb = torch.zeros(ops.norm_tail_ws_doubles(G, Cout), dtype=torch.float64, device="cuda")
    ops.conv3d_fwd_stats(x, None, w, b, y1, wsb, per, act=0)
    torch.cuda.synchronize()
    ws = wsb[:G * Cout * 2].view(G, Cout, 2)             # the totals; behind them the per-workgroup partial sums, folded and cleared
    assert torch.equal(y0, y1) and float(wsb[G * Cout * 2:].abs().max()) == 0.0
    yd = y0.double().reshape(G, -1, Cout)
    ref = torch.stack([yd.sum(1), (yd * yd).sum(1)], dim=-1)
    scale = torch.stack([yd.abs().sum(1), (yd * yd).sum(1)], dim=-1)
    err = float(((ws - ref).abs() / scale).max())
    print("sums vs fp64: max error relative to the sum of magnitudes %.2e" % err)
    assert err <= 2e-6            # fp32 partial sums of <= 128 values per lane, fp64 from there on


@pytest.mark.parametrize("per", [0, 1])
def test_upcat_fwd_stats_tail(ops, per):
    N, D, H, W, C0, C1, Cout = 2, 16, 64, 64, 64, 32, 128
    bf = torch.bfloat16
    assert ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, bf) & 1 and ops.conv3d_fwd_ntail_ok(C1, 0, Cout, N, D, H, W, bf)
    xl, xs = rnd((N, D // 2, H // 2, W // 2, C0), 310, bf), rnd((N, D, H, W, C1), 311, bf)
    w = rnd((27, Cout, C0 + C1), 312, torch.float32, scale=0.05)
    up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=bf)
    sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=bf)
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, None, sk_f, None)
    b = rnd((Cout,), 313, torch.float32)
    y0, y1 = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda"), torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_upcat_fwd(xl, xs, up_f, sk_f, b, y0, act=0)
    G = N if per else 1
    wsb = torch.zeros(ops.norm_tail_ws_doubles(G, Cout), dtype=torch.float64, device="cuda")
    ops.conv3d_upcat_fwd_stats(xl, xs, up_f, sk_f, b, y1, wsb, per, act=0)
    torch.cuda.synchronize()
    ws = wsb[:G * Cout * 2].view(G, Cout, 2)
    assert torch.equal(y0, y1)
    yd = y0.double().reshape(G, -1, Cout)
    ref = torch.stack([yd.sum(1), (yd * yd).sum(1)], dim=-1)
    scale = torch.stack([yd.abs().sum(1), (yd * yd).sum(1)], dim=-1)
    assert float(((ws - ref).abs() / scale).max()) <= 2e-6


@pytest.mark.parametrize("case", NTAIL_CASES, ids=lambda c: "N%d_%dx%dx%d_%d_%d" % c)
@pytest.mark.parametrize("mode", ["batch", "instance"])
@pytest.mark.parametrize("act", [1, 2])
def test_dgrad_norm_tail_matches_the_separate_passes(ops, case, mode, act):
    """conv -> norm -> activation -> conv, backward through the second conv's input gradient into the first block's normalisation:
    fmri_conv3d_dgrad_norm + fmri_norm_act_bwd_pre against fmri_conv3d_dgrad + fmri_norm_act_bwd_x.  ReLU: dz is the same tensor bit
    for bit and the sums differ by the order of summation only; LeakyReLU: the fused form rounds alpha * dy to bf16 before it is summed and
    used (one rounding more), the two forms then agree to bf16 resolution."""
    N, D, H, W, Cdy, Cx = case                 # the second conv: Cx -> Cdy; its input gradient has Cx channels
    bf = torch.bfloat16
    alpha = 0.2
    per = mode == "instance"
    G = N if per else 1
    assert ops.conv3d_fwd_ntail_ok(Cdy, 0, Cx, N, D, H, W, bf)
    x = rnd((N, D, H, W, Cx), 320, bf, scale=1.3) + 0.2                   # the first block's conv output
    gamma = rnd((Cx,), 321, torch.float32) * 0.5 + 1.0
    beta = rnd((Cx,), 322, torch.float32) * 0.3
    stats = torch.zeros((G, Cx, 3), device="cuda")
    ws = torch.zeros((G, Cx, 2), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    ops.norm_act_fwd(x, gamma, beta, y, stats, ws, per, eps=1e-3, eps_on_std=per, act=act, alpha=alpha)
    dy2 = rnd((N, D, H, W, Cdy), 323, bf)                                   # gradient at the second conv's output
    wd = rnd((27, Cx, Cdy), 324, bf, scale=0.05)
    # separate passes
    g0 = torch.empty_like(x)
    ops.conv3d_dgrad(dy2, wd, g0)
    dx0, dg0, db0 = torch.empty_like(x), torch.zeros(Cx, device="cuda"), torch.zeros(Cx, device="cuda")
    ops.norm_act_bwd(x, None, g0, gamma, stats, dx0, dg0, db0, ws, per, act=act, alpha=alpha, beta=beta)
    # tails
    nss = torch.empty((G, Cx, 2), device="cuda")
    ops.norm_scale_shift(stats, gamma, beta, nss)
    dz = torch.empty_like(x)
    wsb = torch.zeros(ops.norm_tail_ws_doubles(G, Cx), dtype=torch.float64, device="cuda")
    ops.conv3d_dgrad_norm(dy2, wd, x, nss, dz, wsb, per, act=act, alpha=alpha)
    torch.cuda.synchronize()
    ws1 = wsb[:G * Cx * 2].view(G, Cx, 2)
    # dz against its definition on the separate path's tensors: the sign of z from the stored y (same sign as z for both activations)
    want = torch.where(y.float() > 0, g0.float(), g0.float() * (0.0 if act == 1 else alpha)).to(bf)
    assert torch.equal(dz, want), "dz: %d of %d values differ" % (int((dz != want).sum()), dz.numel())
    dzd, xd = dz.double().reshape(G, -1, Cx), x.double().reshape(G, -1, Cx)
    ref = torch.stack([dzd.sum(1), (dzd * xd).sum(1)], dim=-1)
    scale = torch.stack([dzd.abs().sum(1), (dzd * xd).abs().sum(1)], dim=-1)
    assert float(((ws1 - ref).abs() / (scale + 1e-30)).max()) <= 2e-6
    dx1, dg1, db1 = torch.empty_like(x), torch.zeros(Cx, device="cuda"), torch.zeros(Cx, device="cuda")
    ops.norm_act_bwd_pre(x, dz, gamma, stats, dx1, dg1, db1, ws1, per)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    e = (rel(dx1, dx0), rel(dg1, dg0), rel(db1, db0))
    print("dx %.2e dgamma %.2e dbeta %.2e" % e)
    bar = 1e-5 if act == 1 else 3e-3                # ReLU: summation order only (+ a rare last-bit flip of a bf16 dx); LeakyReLU: bf16(alpha * dy)
    assert e[0] <= max(bar, 2e-4) and e[1] <= bar and e[2] <= bar


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("planar", [False, True])
def test_deconv_k2s2_fwd_bwd(ops, dtype, planar):
    N, D, H, W, Cin, Cout = 2, 3, 4, 5, 12, 8
    x = torch.relu(rnd((N, D, H, W, Cin), 90, dtype))
    nt = 4 if planar else 8
    w = rnd((8, Cout, Cin), 91, dtype, scale=0.3)
    b = rnd((Cout,), 92, torch.float32)
    D2 = D if planar else 2 * D
    y = torch.empty((N, D2, 2 * H, 2 * W, Cout), dtype=dtype, device="cuda")
    ops.deconv_fwd(x, w, b, y, planar=planar)
    xr = to_ncdhw(f64(x)).requires_grad_(True)
    wr = f64(w).requires_grad_(True)
    br = f64(b).requires_grad_(True)
    if planar:
        k = wr[:4].reshape(2, 2, Cout, Cin).permute(3, 2, 0, 1)               # (Cin,Cout,2,2)
        yr = F.conv_transpose2d(xr.permute(0, 2, 1, 3, 4).reshape(N * D, Cin, H, W), k, br, stride=2)
        yr = yr.reshape(N, D, Cout, 2 * H, 2 * W).permute(0, 2, 1, 3, 4)
    else:
        k = wr.reshape(2, 2, 2, Cout, Cin).permute(4, 3, 0, 1, 2)             # (Cin,Cout,2,2,2)
        yr = F.conv_transpose3d(xr, k, br, stride=2)
    torch.cuda.synchronize()
    assert_close(y, to_ndhwc(yr.detach()), *TOL[dtype], what="deconv fwd")
    dyfull = rnd((N, D2, 2 * H, 2 * W, Cout + 4), 93, dtype)
    dx = torch.empty_like(x)
    dw = torch.zeros((8, Cout, Cin), device="cuda")
    dbg = torch.zeros(Cout, device="cuda")
    ops.deconv_bwd(x, w, dyfull, dx, dw, dbg, dy_off=4, xmask=x, planar=planar)
    torch.cuda.synchronize()
    yr.backward(to_ncdhw(f64(dyfull)[..., 4:]))
    tolb = (1e-4, 1e-5) if dtype == torch.float32 else (5e-3, 1e-4)
    tolw = (1e-4, 1e-5) if dtype == torch.float32 else (2e-5, 2e-5)
    assert_close(dx, to_ndhwc(xr.grad) * (f64(x) > 0), *tolb, what="deconv dx")
    assert_close(dw[:nt], wr.grad[:nt], *tolw, what="deconv dw")
    assert_close(dbg, br.grad, *tolw, what="deconv db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,s,dims", [(1, 1, (3, 4, 5)), (3, 2, (4, 6, 8)), (3, 2, (5, 7, 6)), (3, 1, (3, 4, 5))])
def test_conv_direct_fwd_bwd(ops, dtype, k, s, dims):
    """kernel 1 / stride 2 convolutions with TensorFlow 'same' padding (even extent: 0 before, 1 after)"""
    N, Cin, Cout = 2, 6, 5
    D, H, W = dims
    x = rnd((N, D, H, W, Cin), 100, dtype)
    w = rnd((k ** 3, Cout, Cin), 101, dtype, scale=0.3)
    b = rnd((Cout,), 102, torch.float32)
    od = [-(-n // s) for n in dims]
    y = torch.empty((N,) + tuple(od) + (Cout,), dtype=dtype, device="cuda")
    ops.conv_direct_fwd(x, w, b, y, k, s, act=0)
    xr = to_ncdhw(f64(x)).requires_grad_(True)
    wr = f64(w).requires_grad_(True)
    br = f64(b).requires_grad_(True)
    pads = []
    for n_, o_ in zip(reversed(dims), reversed(od)):           # F.pad wants (w_before, w_after, h_before, h_after, d_before, d_after)
        tot = max((o_ - 1) * s + k - n_, 0)
        pads += [tot // 2, tot - tot // 2]
    yr = F.conv3d(F.pad(xr, pads), wr.reshape(k, k, k, Cout, Cin).permute(3, 4, 0, 1, 2), br, stride=s)
    torch.cuda.synchronize()
    assert_close(y, to_ndhwc(yr.detach()), *TOL[dtype], what="direct fwd")
    dy = rnd(tuple(y.shape), 103, dtype)
    dx = torch.empty_like(x)
    dw = torch.zeros((k ** 3, Cout, Cin), device="cuda")
    db = torch.zeros(Cout, device="cuda")
    ops.conv_direct_bwd(x, w, dy, dx, dw, db, k, s)
    torch.cuda.synchronize()
    yr.backward(to_ncdhw(f64(dy)))
    tolb = (1e-4, 1e-5) if dtype == torch.float32 else (5e-3, 1e-4)
    tolw = (1e-4, 1e-5) if dtype == torch.float32 else (2e-5, 2e-5)
    assert_close(dx, to_ndhwc(xr.grad), *tolb, what="direct dx")
    assert_close(dw, wr.grad, *tolw, what="direct dw")
    assert_close(db, br.grad, *tolw, what="direct db")


def test_add_and_channel_scale(ops):
    a, b = rnd((2, 3, 4, 5, 8), 110, torch.bfloat16), rnd((2, 3, 4, 5, 8), 111, torch.bfloat16)
    y = torch.empty_like(a)
    ops.add(a, b, y)
    sc = (torch.rand(2, 8, generator=torch.Generator().manual_seed(1)) > 0.3).float().cuda() / 0.7
    z = torch.empty_like(a)
    ops.channel_scale(a, sc, z)
    torch.cuda.synchronize()
    assert torch.equal(y, (a.float() + b.float()).to(torch.bfloat16))
    assert torch.equal(z, (a.float() * sc[:, None, None, None, :]).to(torch.bfloat16))


@pytest.mark.parametrize("name,kind,param", [("dice_coefficient_loss", 0, 1.0), ("binary_crossentropy_loss", 1, 1.0), ("dice_and_xent", 2, 1.0),
                                             ("focal_loss", 3, 1.0), ("vod_coefficient_loss", 4, 1.0), ("double_dice_loss", 5, 10.0)])
def test_selectable_losses_value_and_gradient(ops, name, kind, param):
    """the loss table of reference fetal/config_utils.py / fetal_net/metrics.py: value (vs the pinned numpy restatement) and d/dlogits
    (vs autograd of the same formula)"""
    from oracle import metrics_oracle as M
    n = 4000
    logits = rnd((n,), 120, torch.float32, scale=2.0)
    y = (torch.rand(n, generator=torch.Generator().manual_seed(121)) > 0.7).to(torch.uint8).cuda()
    probs = torch.empty_like(logits)
    sums = torch.zeros(16, dtype=torch.float64, device="cuda")
    ops.sigmoid_dice_fwd(logits, y, probs, sums)
    dl = torch.empty_like(logits)
    ops.sigmoid_loss_bwd(probs, y, sums, dl, kind, param)
    torch.cuda.synchronize()
    yn, pn = y.cpu().numpy().astype(np.float64), torch.sigmoid(f64(logits)).numpy()
    ref_val = {"dice_coefficient_loss": M.dice_coefficient_loss, "binary_crossentropy_loss": lambda a, b: M.weighted_cross_entropy_loss(a, b),
               "dice_and_xent": M.dice_and_xent, "focal_loss": M.focal_loss, "vod_coefficient_loss": M.vod_coefficient_loss,
               "double_dice_loss": M.double_dice_loss}[name](yn, pn)
    assert ops.loss_value_from_sums(sums.cpu().numpy(), kind, param) == pytest.approx(ref_val, rel=2e-5)
    z = f64(logits).requires_grad_(True)
    p = torch.sigmoid(z)
    t = f64(y)
    dice = lambda a, b: (2 * (a * b).sum() + 1) / (a.sum() + b.sum() + 1)
    pc = p.clamp(1e-7, 1 - 1e-7)
    xent = -(t * torch.log(pc) + (1 - t) * torch.log(1 - pc)).mean()
    L = {0: -dice(t, p), 1: xent, 2: -dice(t, p) + param * xent,
         3: -(0.5 * (1 - p) ** 2 * torch.log(p))[t == 1].sum() - (0.5 * p ** 2 * torch.log(1 - p))[t == 0].sum(),
         4: -((t * p).sum() + 1) / (t.sum() + p.sum() - (t * p).sum() + 1), 5: -dice(t, p) + param * dice(1 - t, p)}[kind]
    L.backward()
    assert_close(dl, z.grad, 2e-4, 2e-5, what=name)


# ---------------------------------------------------------------------------------------------- up-sample + concat + conv, parity form
UPCAT_CASES = [
    # name, N, D,H,W (output dims), C0 (up-sampled), C1 (skip), Cout
    ("one_tile", 1, 8, 16, 32, 64, 32, 64),
    ("multi_tile", 2, 16, 32, 64, 128, 64, 64),
    ("narrow_cout", 1, 8, 32, 32, 64, 64, 96),
    ("deep", 1, 8, 16, 32, 256, 128, 128),
]


@pytest.mark.parametrize("case", UPCAT_CASES, ids=[c[0] for c in UPCAT_CASES])
def test_upcat_fwd_and_dgrad(ops, case):
    """fmri_conv3d_upcat_fwd / _dgrad (8 parity classes x 8 pre-summed taps on the low-res tensor) against the plain definition
    conv3x3x3(concat(nearest_up2(x_low), x_skip)) and its autograd gradients in fp64.  Inputs and master weights are bf16-exact;
    the parity form rounds the pre-summed weights and the up-sampled channels' partial sum to bf16 once more, hence PARITY_TOL."""
    name, N, D, H, W, C0, C1, Cout = case
    dtype = torch.bfloat16
    assert ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, dtype) & 1
    x_low = rnd((N, D // 2, H // 2, W // 2, C0), 1, dtype)
    x_skip = rnd((N, D, H, W, C1), 2, dtype)
    w = rnd((27, Cout, C0 + C1), 3, dtype, scale=0.05).float().contiguous()            # fp32 master holding bf16-exact values
    bias = rnd((Cout,), 4, torch.float32)
    up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=dtype)
    up_d = torch.empty((8, 8, C0, Cout), device="cuda", dtype=dtype)
    sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=dtype)
    sk_d = torch.empty((27, C1, Cout), device="cuda", dtype=dtype)
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_upcat_fwd(x_low, x_skip, up_f, sk_f, bias, y, act=1)
    torch.cuda.synchronize()
    # checker: plain definition with autograd
    xl = f64(x_low).requires_grad_(True)
    xs = f64(x_skip).requires_grad_(True)
    inp = ref_concat_input(xl, xs, True)
    pre = F.conv3d(inp, keras_kernel_from_packed(f64(w)), f64(bias), padding=1)
    ref = to_ndhwc(F.relu(pre))
    assert_close(y, ref, *PARITY_TOL, what=name + " fwd")
    # the existing fused-upsample kernel computes the same thing with 27 taps: the two device paths agree as well
    wf = w.to(dtype)
    y2 = torch.empty_like(y)
    ops.conv3d_fwd(x_low, x_skip, wf, bias, y2, up0=True, act=1)
    assert_close(y, f64(y2), *PARITY_TOL, what=name + " fwd vs 27-tap kernel")
    # gradients w.r.t. both inputs for a random dy, with the producers' ReLU masks
    dy = rnd((N, D, H, W, Cout), 5, dtype)
    m_low = rnd((N, D // 2, H // 2, W // 2, C0), 6, dtype).clamp_min(0)
    m_skip = rnd((N, D, H, W, C1), 7, dtype).clamp_min(0)
    dx_low = torch.full_like(x_low, float("nan"))
    dx_skip = torch.full_like(x_skip, float("nan"))
    ops.conv3d_upcat_dgrad(dy, up_d, sk_d, m_low, m_skip, dx_low, dx_skip)
    torch.cuda.synchronize()
    pre.backward(to_ncdhw(f64(dy)))
    assert_close(dx_low, xl.grad * (f64(m_low) > 0), *PARITY_TOL, what=name + " dx_low")
    assert_close(dx_skip, xs.grad * (f64(m_skip) > 0), *TOL[torch.bfloat16], what=name + " dx_skip")
    # weight / bias gradient: parity form vs autograd of the plain definition (fp32 accumulation of bf16 products: 2e-3)
    if ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, dtype) & 2:
        dw = torch.zeros((27, Cout, C0 + C1), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        scratch = torch.empty(64 * Cout * C0, device="cuda")
        ops.conv3d_upcat_wgrad(x_low, x_skip, dy, dw, db, scratch)
        torch.cuda.synchronize()
        kern = keras_kernel_from_packed(f64(w)).requires_grad_(True)
        pre2 = F.conv3d(ref_concat_input(f64(x_low), f64(x_skip), True), kern, f64(bias), padding=1)
        pre2.backward(to_ncdhw(f64(dy)))
        ref_dw = kern.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
        assert_close(dw, ref_dw, 4e-6, 4e-6, what=name + " dw")
        assert_close(db, f64(dy).sum(dim=(0, 1, 2, 3)), 4e-6, 4e-6, what=name + " db")


def test_upcat_without_skip_channels(ops):
    """C1 = 0: convolution of a purely up-sampled tensor (the Isensee up-sampling module): one parity launch with bias + LeakyReLU in
    its own epilogue; gradients incl. the bias gradient taken inside the parity weight-gradient kernel"""
    N, D, H, W, C0, Cout = 1, 8, 16, 32, 64, 64
    dtype = torch.bfloat16
    assert ops.conv3d_upcat_ok(C0, 0, Cout, D, H, W, dtype) == 3
    x_low = rnd((N, D // 2, H // 2, W // 2, C0), 11, dtype)
    w = rnd((27, Cout, C0), 12, dtype, scale=0.05).float().contiguous()
    bias = rnd((Cout,), 13, torch.float32)
    up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=dtype)
    up_d = torch.empty((8, 8, C0, Cout), device="cuda", dtype=dtype)
    ops.conv3d_pack_up_weights(w, C0, 0, up_f, up_d, None, None)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_upcat_fwd(x_low, None, up_f, None, bias, y, act=2, alpha=0.3)
    torch.cuda.synchronize()
    xl = f64(x_low).requires_grad_(True)
    kern = keras_kernel_from_packed(f64(w)).requires_grad_(True)
    b64 = f64(bias).requires_grad_(True)
    pre = F.conv3d(ref_concat_input(xl, None, True), kern, b64, padding=1)
    assert_close(y, to_ndhwc(F.leaky_relu(pre, 0.3)), *PARITY_TOL, what="fwd")
    dy = rnd((N, D, H, W, Cout), 14, dtype)
    dx_low = torch.full_like(x_low, float("nan"))
    ops.conv3d_upcat_dgrad(dy, up_d, None, None, None, dx_low, None)
    dw, db = torch.zeros((27, Cout, C0), device="cuda"), torch.zeros(Cout, device="cuda")
    ops.conv3d_upcat_wgrad(x_low, None, dy, dw, db, torch.empty(64 * Cout * C0, device="cuda"))
    torch.cuda.synchronize()
    pre.backward(to_ncdhw(f64(dy)))
    assert_close(dx_low, xl.grad, *PARITY_TOL, what="dx_low")
    assert_close(dw, kern.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0), 4e-6, 4e-6, what="dw")
    assert_close(db, b64.grad, 4e-6, 4e-6, what="db")


UPCAT2D_CASES = [
    # name, S (slices), H, W (output dims), C0 (up-sampled), C1 (skip), Cout
    ("one_tile", 4, 16, 32, 64, 32, 64),
    ("multi_tile", 8, 32, 64, 128, 64, 64),
    ("narrow_cout", 4, 32, 32, 64, 64, 96),
    ("deep", 4, 16, 32, 256, 128, 128),
    ("no_skip", 4, 16, 32, 64, 0, 64),
]


@pytest.mark.parametrize("case", UPCAT2D_CASES, ids=[c[0] for c in UPCAT2D_CASES])
def test_upcat_2d_fwd_dgrad_wgrad(ops, case):
    """fmri_conv2d_upcat_* (4 parity classes x 4 pre-summed taps per slice) against conv3x3(concat(nearest_up2(x_low), x_skip)) per slice
    and its fp64 autograd gradients (reference model/unet/unet.py:60-66).  Tensors [1][S][H][W][C]; the 3x3 kernel is the centre kd plane
    of the 27-tap weight image, the other planes are poisoned to prove they are never read."""
    name, S, H, W, C0, C1, Cout = case
    dtype = torch.bfloat16
    ok = ops.conv3d_upcat_ok(C0, C1, Cout, S, H, W, dtype, planar=True)
    assert ok & 1
    x_low = rnd((1, S, H // 2, W // 2, C0), 21, dtype)
    x_skip = rnd((1, S, H, W, C1), 22, dtype) if C1 else None
    w = rnd((27, Cout, C0 + C1), 23, dtype, scale=0.05).float().contiguous()
    w[:9] = 1e4
    w[18:] = -1e4
    bias = rnd((Cout,), 24, torch.float32)
    up_f = torch.empty((4, 4, Cout, C0), device="cuda", dtype=dtype)
    up_d = torch.empty((4, 4, C0, Cout), device="cuda", dtype=dtype)
    sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=dtype) if C1 else None
    sk_d = torch.empty((27, C1, Cout), device="cuda", dtype=dtype) if C1 else None
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d, planar=True)
    y = torch.full((1, S, H, W, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.conv3d_upcat_fwd(x_low, x_skip, up_f, sk_f, bias, y, act=1, planar=True)
    torch.cuda.synchronize()

    def ref_pre(xl, xs, kern, b):
        up = xl[0].permute(0, 3, 1, 2)                                            # [S][C0][h][w]
        up = up.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        inp = up if xs is None else torch.cat([up, xs[0].permute(0, 3, 1, 2)], dim=1)
        return F.conv2d(inp, kern, b, padding=1)                                  # [S][Cout][H][W]

    xl = f64(x_low).requires_grad_(True)
    xs = f64(x_skip).requires_grad_(True) if C1 else None
    kern = f64(w)[9:18].reshape(3, 3, Cout, C0 + C1).permute(2, 3, 0, 1).contiguous().requires_grad_(True)
    b64 = f64(bias).requires_grad_(True)
    pre = ref_pre(xl, xs, kern, b64)
    ref = F.relu(pre).permute(0, 2, 3, 1).unsqueeze(0)
    assert_close(y, ref, *PARITY_TOL, what=name + " fwd")
    dy = rnd((1, S, H, W, Cout), 25, dtype)
    m_low = rnd((1, S, H // 2, W // 2, C0), 26, dtype).clamp_min(0)
    m_skip = rnd((1, S, H, W, C1), 27, dtype).clamp_min(0) if C1 else None
    dx_low = torch.full_like(x_low, float("nan"))
    dx_skip = torch.full_like(x_skip, float("nan")) if C1 else None
    ops.conv3d_upcat_dgrad(dy, up_d, sk_d, m_low, m_skip, dx_low, dx_skip, planar=True)
    torch.cuda.synchronize()
    pre.backward(f64(dy)[0].permute(0, 3, 1, 2))
    assert_close(dx_low, xl.grad * (f64(m_low) > 0), *PARITY_TOL, what=name + " dx_low")
    if C1:
        assert_close(dx_skip, xs.grad * (f64(m_skip) > 0), *TOL[torch.bfloat16], what=name + " dx_skip")
    if ok & 2:
        dw = torch.zeros((27, Cout, C0 + C1), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        ops.conv3d_upcat_wgrad(x_low, x_skip, dy, dw, db, torch.empty(16 * Cout * C0, device="cuda"), planar=True)
        torch.cuda.synchronize()
        ref_dw = kern.grad.permute(2, 3, 0, 1).reshape(9, Cout, C0 + C1)
        assert_close(dw[9:18], ref_dw, 4e-6, 4e-6, what=name + " dw")
        assert float(dw[:9].abs().max()) == 0.0 and float(dw[18:].abs().max()) == 0.0
        assert_close(db, b64.grad, 4e-6, 4e-6, what=name + " db")
    else:
        assert Cout % 64                                                          # the weight-gradient kernel tiles Cout by 64


def test_weighted_cross_entropy_loss_value_and_gradient(ops):
    """fmri_sigmoid_dice_fwd_weighted / _loss_bwd_weighted = Dice + w * mean(exp(-mask/sigma) * BCE) (reference metrics.py:66-76,89-95)
    against the metrics oracle's dice_and_xent with a weight mask, value and finite-difference-free analytic gradient (torch fp64)."""
    from oracle import metrics_oracle as MO
    rs = np.random.RandomState(3)
    n = 4096
    logits = torch.tensor(rs.randn(n) * 2, dtype=torch.float32, device="cuda")
    y = torch.tensor((rs.rand(n) > 0.6).astype(np.uint8), device="cuda")
    mask = torch.tensor(rs.rand(n) * 9, dtype=torch.float32, device="cuda")
    weight = torch.exp(-mask / 3.0)
    probs, sums, dl = torch.empty_like(logits), torch.zeros(16, dtype=torch.float64, device="cuda"), torch.empty_like(logits)
    ops.sigmoid_dice_fwd(logits, y, probs, sums, weight=weight)
    ops.sigmoid_loss_bwd(probs, y, sums, dl, 2, param=0.7, weight=weight)
    torch.cuda.synchronize()
    got = ops.loss_value_from_sums(sums.cpu().numpy(), 2, 0.7)
    z = logits.cpu().double().requires_grad_(True)
    p = torch.sigmoid(z)
    t = y.cpu().double()
    w64 = weight.cpu().double()
    dice = (2 * (t * p).sum() + 1) / (t.sum() + p.sum() + 1)
    pc = p.clamp(1e-7, 1 - 1e-7)
    loss = -dice + 0.7 * (w64 * -(t * torch.log(pc) + (1 - t) * torch.log(1 - pc))).mean()
    loss.backward()
    want = MO.dice_and_xent(t.numpy(), p.detach().numpy(), xent_weight=0.7, weight_mask=w64.numpy()) if hasattr(MO, "dice_and_xent") else float(loss.detach())
    assert abs(got - float(loss.detach())) <= 1e-5 and abs(got - float(want)) <= 1e-5
    assert_close(dl, z.grad, 1e-4, 1e-5, what="weighted dice+xent gradient")


TAIL_CASES = [
    # name, N, D, H, W, C0, Cout, expected ok bits (on a 256-CU device)
    ("narrow_32", 1, 8, 16, 32, 32, 32, 3),          # 32-wide Cout block == Cout: pool + logits
    ("few_tiles_64", 2, 8, 16, 32, 32, 64, 1),       # fewer (tile, 64-block) pairs than CUs -> 32-wide blocks: no single block sees all 64 channels
    ("wide_64", 2, 16, 64, 64, 32, 64, 3),           # the shape class of the network's last block (64-wide block == Cout)
    ("wide_128_pool", 1, 16, 64, 128, 64, 128, 1),   # two 64-blocks per voxel: pool only
]


@pytest.mark.parametrize("case", TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_conv3d_fwd_tail(ops, case):
    """fmri_conv3d_fwd_tail: the conv block's epilogue also writes MaxPooling3D(2)(y) and the final 1x1x1 conv's logits from the tile
    in LDS.  y must be BIT-identical to fmri_conv3d_fwd, the pooled tensor to fmri_maxpool3d_2x_fwd(y) (same bf16 values, max is
    exact), the logits equal fmri_conv1x1_fwd(y) up to fp32 summation order; all three are checked against fp64 as well
    (reference unet3d/unet.py:45-51, :68)."""
    name, N, D, H, W, C0, Cout, bits = case
    bf = torch.bfloat16
    ok = ops.conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, bf)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert ok == bits, (name, ok)
    x = rnd((N, D, H, W, C0), 21, bf)
    w = rnd((27, Cout, C0), 22, bf, scale=0.1)
    bias = rnd((Cout,), 23, torch.float32, scale=0.3)
    w1 = rnd((Cout,), 24, torch.float32, scale=0.2)
    b1 = rnd((1,), 25, torch.float32)
    y0 = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x, None, w, bias, y0, act=1)
    p0 = torch.empty((N, D // 2, H // 2, W // 2, Cout), dtype=bf, device="cuda")
    ops.maxpool_fwd(y0, p0)
    l0 = torch.empty((N * D * H * W, 1), dtype=torch.float32, device="cuda")
    ops.conv1x1_fwd(y0, w1.reshape(1, Cout), b1, l0)
    y = torch.full_like(y0, float("nan"))
    pool = torch.full_like(p0, float("nan")) if ok & 1 else None
    logits = torch.full((N * D * H * W,), float("nan"), dtype=torch.float32, device="cuda") if ok & 2 else None
    assert ok & 1
    ops.conv3d_fwd_tail(x, w, bias, y, pool=pool, w1=w1 if ok & 2 else None, b1=b1 if ok & 2 else None, logits=logits, act=1)
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y0.view(torch.int16))
    assert torch.equal(pool.view(torch.int16), p0.view(torch.int16))
    ref = ref_conv_fwd(f64(x), None, False, f64(w), f64(bias), 1)
    assert_close(y, ref, *TOL[bf], what=name + " tail y")
    if ok & 2:
        assert_close(logits.reshape(-1, 1), l0, 2e-6, 2e-6, what=name + " tail logits vs conv1x1 kernel")
        assert_close(logits, ((f64(y) @ f64(w1)) + f64(b1)).reshape(-1), 2e-6, 2e-6, what=name + " tail logits vs fp64")
    else:
        with pytest.raises(RuntimeError):
            ops.conv3d_fwd_tail(x, w, bias, y, w1=w1, b1=b1, logits=torch.empty(N * D * H * W, dtype=torch.float32, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_pack_weights_batched_equals_the_per_layer_launches(ops, dtype):
    """fmri_pack_weights_batched: every weight image of a model in one launch - plain layers (ragged channel counts included), a 3-D and a
    2-D parity-form layer, destinations that are not wanted - against fmri_conv3d_pack_weights / fmri_conv3d|conv2d_pack_up_weights: the
    same bits (a workgroup runs the layer's own code on its local block number)."""
    g = torch.Generator().manual_seed(17)
    mk = lambda *sh: torch.randn(*sh, generator=g).cuda()
    em = lambda *sh: torch.full(sh, float("nan"), dtype=dtype, device="cuda")
    plain = [(64, 32), (128, 64), (40, 24), (96, 200)]                       # (Cout, Cin): tiles that do not fill 64 x 64 as well
    ups = [(128, 64, 64, False), (64, 32, 32, True), (96, 0, 64, False)]     # (C0, C1, Cout, planar); C1 = 0: purely up-sampled input
    ent, ref, got = [], [], []
    for i, (co, ci) in enumerate(plain):
        w = mk(27, co, ci)
        a = [em(27, co, ci), em(27, ci, co) if i != 1 else None]             # one layer without the input-gradient image
        b = [em(27, co, ci), em(27, ci, co) if i != 1 else None]
        ops.pack_weights(w, a[0], a[1])
        ent.append(("plain", w, b[0], b[1]))
        ref += a
        got += b
    for c0, c1, co, planar in ups:
        w = mk(27, co, c0 + c1)
        n = 4 if planar else 8
        mkimg = lambda: [em(n, n, co, c0), em(n, n, c0, co), em(27, co, c1) if c1 else None, em(27, c1, co) if c1 else None]
        a, b = mkimg(), mkimg()
        ops.conv3d_pack_up_weights(w, c0, c1, a[0], a[1], a[2], a[3], planar=planar)
        ent.append(("up", w, c0, c1, b[0], b[1], b[2], b[3], planar))
        ref += a
        got += b
    tab, nb = ops.pack_table(ent, "cuda")
    ops.pack_weights_batched(tab, nb, dtype)
    torch.cuda.synchronize()
    view = lambda t: t.view(torch.int16 if dtype == torch.bfloat16 else torch.int32)
    for k, (r, o) in enumerate(zip(ref, got)):
        if r is not None:
            assert torch.equal(view(r), view(o)), k


def test_engine_batched_repack_equals_the_per_layer_repack(monkeypatch):
    """the engine's weight images after an optimizer step with FMRI_PACK_BATCHED=1 (default) and =0: the same bits, 3-D and 2-D"""
    from fmri_hip.engine import UNetEngine, UNetPlan
    for ndim in (3, 2):
        imgs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("FMRI_PACK_BATCHED", mode)
            if ndim == 3:
                eng = UNetEngine(UNetPlan(1, (16, 32, 32), depth=3, n_base_filters=32), 1, dtype=torch.bfloat16, seed=5)
                x = torch.randn((1, 16, 32, 32, 1), generator=torch.Generator().manual_seed(1)).cuda().to(torch.bfloat16)
                nv = 16 * 32 * 32
            else:
                eng = UNetEngine(UNetPlan(5, (64, 64), depth=3, n_base_filters=32, ndim=2), 8, dtype=torch.bfloat16, seed=5)
                x = torch.randn((1, 8, 64, 64, 5), generator=torch.Generator().manual_seed(1)).cuda().to(torch.bfloat16)
                nv = 8 * 64 * 64
            y = (torch.rand((nv,), generator=torch.Generator().manual_seed(2)) > 0.7).to(torch.uint8).cuda()
            eng.P.copy_(torch.randn(eng.P.shape, generator=torch.Generator().manual_seed(3)) * 0.05)     # the same parameters in both arms
            eng.refresh_weight_copies(overlap=True)
            eng._join_packs()
            torch.cuda.synchronize()
            imgs[mode] = {("f", k): v.clone() for k, v in eng.Wf.items()}
            imgs[mode].update({("d", k): v.clone() for k, v in eng.Wd.items()})
            for k, W in eng.Wup.items():
                imgs[mode].update({(kk, k): v.clone() for kk, v in W.items() if v is not None})
        assert imgs["1"].keys() == imgs["0"].keys() and len(imgs["1"]) > 10
        for k in imgs["1"]:
            assert torch.equal(imgs["1"][k].view(torch.int16), imgs["0"][k].view(torch.int16)), (ndim, k)


PLANAR_TAIL_CASES = [
    # name, slices, H, W, C0, Cout, act, expected ok bits (on a 256-CU device)
    ("narrow_32", 8, 16, 32, 32, 32, 1, 3),             # 32-wide Cout block == Cout: pool + logits
    ("few_tiles_64", 8, 32, 32, 32, 64, 1, 1),          # fewer (tile, 64-block) pairs than CUs -> 32-wide blocks: pool only
    ("wide_64", 16, 64, 128, 32, 64, 1, 3),             # the shape class of the 2-D network's last block
    ("wide_64_two_chunks", 16, 64, 128, 64, 64, 1, 3),
    ("wide_128_pool", 8, 64, 128, 64, 128, 1, 1),       # two 64-blocks per voxel: pool only
    ("wide_64_leaky", 16, 64, 128, 32, 64, 2, 3),       # not ReLU: the pooled maximum goes through fp32 (negative values)
    ("wide_64_linear", 16, 64, 128, 32, 64, 0, 3),
]


@pytest.mark.parametrize("case", PLANAR_TAIL_CASES, ids=[c[0] for c in PLANAR_TAIL_CASES])
def test_conv3d_fwd_tail_planar(ops, case):
    """fmri_conv3d_fwd_tail_planar (2-D models): the planar conv block's epilogue also writes MaxPooling2D(2) of every slice and the final
    1x1 conv's logits from the staged tile.  y BIT-identical to fmri_conv3d_fwd(planar), the pooled slices to fmri_maxpool3d_2x_fwd(planar),
    the logits equal to fmri_conv1x1_fwd up to fp32 summation order; y and logits against fp64 as well (reference unet/unet.py:57-67, :82)."""
    name, S, H, W, C0, Cout, act, bits = case
    bf = torch.bfloat16
    ok = ops.conv3d_fwd_tail_ok(C0, Cout, 1, S, H, W, bf, planar=True)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert ok == bits, (name, ok)
    x = rnd((1, S, H, W, C0), 31, bf)
    w = rnd((27, Cout, C0), 32, bf, scale=0.1)
    bias = rnd((Cout,), 33, torch.float32, scale=0.3)
    w1 = rnd((Cout,), 34, torch.float32, scale=0.2)
    b1 = rnd((1,), 35, torch.float32)
    y0 = torch.empty((1, S, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x, None, w, bias, y0, act=act, alpha=0.1, planar=True)
    p0 = torch.empty((1, S, H // 2, W // 2, Cout), dtype=bf, device="cuda")
    ops.maxpool_fwd(y0, p0, planar=True)
    l0 = torch.empty((S * H * W, 1), dtype=torch.float32, device="cuda")
    ops.conv1x1_fwd(y0, w1.reshape(1, Cout), b1, l0)
    y = torch.full_like(y0, float("nan"))
    pool = torch.full_like(p0, float("nan"))
    logits = torch.full((S * H * W,), float("nan"), dtype=torch.float32, device="cuda") if ok & 2 else None
    assert ok & 1
    ops.conv3d_fwd_tail(x, w, bias, y, pool=pool, w1=w1 if ok & 2 else None, b1=b1 if ok & 2 else None, logits=logits, act=act, alpha=0.1,
                        planar=True)
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y0.view(torch.int16))
    assert torch.equal(pool.view(torch.int16), p0.view(torch.int16))
    ref = ref_conv_fwd(f64(x), None, False, f64(w), f64(bias), 1 if act == 1 else 0, planar=True)
    if act == 2:
        ref = torch.where(ref > 0, ref, 0.1 * ref)
    assert_close(y, ref, *TOL[bf], what=name + " planar tail y")
    if ok & 2:
        assert_close(logits.reshape(-1, 1), l0, 2e-6, 2e-6, what=name + " planar tail logits vs conv1x1 kernel")
        assert_close(logits, ((f64(y) @ f64(w1)) + f64(b1)).reshape(-1), 2e-6, 2e-6, what=name + " planar tail logits vs fp64")
        # the logits alone (no pooled copy)
        lg = torch.full_like(logits, float("nan"))
        ops.conv3d_fwd_tail(x, w, bias, y, w1=w1, b1=b1, logits=lg, act=act, alpha=0.1, planar=True)
        assert torch.equal(lg, logits)
    else:
        with pytest.raises(RuntimeError):
            ops.conv3d_fwd_tail(x, w, bias, y, w1=w1, b1=b1, logits=torch.empty(S * H * W, dtype=torch.float32, device="cuda"), planar=True)


def test_engine_tail_fusion_equals_separate_kernels(monkeypatch):
    """the engine with the pooling / final-conv tails fused into the conv epilogues (default) against FMRI_TAIL_FUSE=0: identical
    activations and pooled tensors, logits equal to fp32 summation order, same Dice, gradients equal to atomics order"""
    from fmri_hip.engine import UNetEngine, UNetPlan
    sp, N = (16, 64, 64), 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N,) + sp + (1,), generator=g).cuda().to(torch.bfloat16)
    yv = (torch.rand((N * sp[0] * sp[1] * sp[2],), generator=g) > 0.7).to(torch.uint8).cuda()
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("FMRI_TAIL_FUSE", fuse)
        eng = UNetEngine(UNetPlan(1, sp, depth=3, n_base_filters=32), N, dtype=torch.bfloat16, seed=11)
        if fuse == "1":
            assert eng._tail_ok(eng.plan.enc[0][1]) & 1 and eng._tail_ok(eng.plan.dec[-1][1]) & 2
        eng.forward(x)
        s = eng.loss_forward(yv).cpu().numpy().copy()
        eng.backward(yv)
        torch.cuda.synchronize()
        out[fuse] = (eng.logits.clone(), {k: v.clone() for k, v in eng.act.items()}, s, eng.G.clone())
    la, aa, sa, ga = out["1"]
    lb, ab, sb, gb = out["0"]
    for k in aa:
        assert torch.equal(aa[k].view(torch.int16), ab[k].view(torch.int16)), k
    assert_close(la, lb, 2e-6, 2e-6, what="fused logits")
    assert abs(sa[0] - sb[0]) <= 1e-6 * abs(sb[0]) and sa[7] == sb[7]
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())


@pytest.mark.parametrize("slices", [16, 6])
def test_engine_2d_tail_fusion_equals_separate_kernels(monkeypatch, slices):
    """the 2-D engine with MaxPooling2D / the final Conv2D in the planar conv epilogues (default) against FMRI_TAIL_FUSE_2D=0: identical
    activations and pooled tensors, logits equal to fp32 summation order, same Dice, gradients equal to atomics order.  6 slices: not a
    multiple of the 4-slice tile - nothing is fused there and the engine must still agree with itself."""
    from fmri_hip.engine import UNetEngine, UNetPlan
    X, Y, C = 64, 128, 5
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, slices, X, Y, C), generator=g).cuda().to(torch.bfloat16)
    yv = (torch.rand((slices * X * Y,), generator=g) > 0.7).to(torch.uint8).cuda()
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("FMRI_TAIL_FUSE_2D", fuse)
        eng = UNetEngine(UNetPlan(C, (X, Y), depth=3, n_base_filters=32, ndim=2), slices, dtype=torch.bfloat16, seed=11)
        if fuse == "1" and slices % 4 == 0:
            assert eng._tail_ok(eng.plan.enc[0][1]) & 1 and eng._tail_ok(eng.plan.dec[-1][1]) & 2
        if fuse == "0":
            assert eng._tail_ok(eng.plan.enc[0][1]) == 0
        eng.forward(x)
        s = eng.loss_forward(yv).cpu().numpy().copy()
        eng.backward(yv)
        torch.cuda.synchronize()
        out[fuse] = (eng.logits.clone(), {k: v.clone() for k, v in eng.act.items()}, s, eng.G.clone())
    la, aa, sa, ga = out["1"]
    lb, ab, sb, gb = out["0"]
    for k in aa:
        assert torch.equal(aa[k].view(torch.int16), ab[k].view(torch.int16)), k
    assert_close(la, lb, 2e-6, 2e-6, what="fused 2-D logits")
    assert abs(sa[0] - sb[0]) <= 1e-6 * abs(sb[0]) and sa[7] == sb[7]
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())


# ------------------------------------------------------------------------------------------------ seeded fuzz of the dispatch
def _fuzz_cases(n=14, seed=2024):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        N, D, H, W = int(rs.choice([1, 2, 3, 5])), int(rs.choice([4, 8, 12, 20])), int(rs.choice([8, 16, 24, 40])), int(rs.choice([16, 32, 48]))
        if N * D * H * W > 70000:
            continue
        C0, C1, Cout = int(rs.choice([32, 64, 96])), int(rs.choice([0, 0, 32, 64])), int(rs.choice([32, 64, 128]))
        up0 = bool(rs.rand() < 0.3)
        out.append((N, D, H, W, C0, up0, C1, Cout))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "N%d_%dx%dx%d_c%d%s+%d_o%d" % (c[0], c[1], c[2], c[3], c[4], "up" if c[5] else "", c[6], c[7]))
def test_conv_dispatch_fuzz_is_exact_on_dyadic_data(ops, case):
    """random grids (odd tile counts, batch 3 and 5, grids that do and do not allow the compact XCD numbering) through the AUTO dispatch:
    forward, input gradient and weight gradient on small dyadic values, where every product and every partial sum is exact in fp32 - the
    bf16 result must equal the rounded CPU result bit for bit, the fp32 weight gradient exactly"""
    N, D, H, W, C0, up0, C1, Cout = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(N * 1000 + D + H + W + C0 + Cout)
    dy4 = lambda shape, lo=-4, hi=5, div=4.0: (torch.randint(lo, hi, shape, generator=g).float() / div)
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    x0 = dy4(s0)
    x1 = dy4((N, D, H, W, C1)) if C1 else None
    w = dy4((27, Cout, C0 + C1), -2, 3, 8.0)
    bias = dy4((Cout,))
    xin = ref_concat_input(x0, x1, up0).float()                         # NCDHW fp32 on the CPU
    wk = keras_kernel_from_packed(w).float().requires_grad_(True)
    xin.requires_grad_(True)
    yref = F.conv3d(xin, wk, bias, padding=1)
    y = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x0.to(bf).cuda(), None if x1 is None else x1.to(bf).cuda(), w.to(bf).cuda(), bias.cuda(), y, up0=up0, act=1)
    want = to_ndhwc(F.relu(yref).detach()).to(bf)
    assert torch.equal(y.cpu().view(torch.int16), want.view(torch.int16)), "forward"
    # weight gradient (fp32 sums of exact products: order-independent) and bias gradient
    dy = dy4((N, D, H, W, Cout), -2, 3, 2.0)
    yref.backward(to_ncdhw(dy).float())
    dw = torch.zeros((27, Cout, C0 + C1), dtype=torch.float32, device="cuda")
    db = torch.zeros((Cout,), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(x0.to(bf).cuda(), None if x1 is None else x1.to(bf).cuda(), dy.to(bf).cuda(), dw, db, up0=up0)
    ref_dw = wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
    assert torch.equal(dw.cpu(), ref_dw), "weight gradient: max diff %g" % float((dw.cpu() - ref_dw).abs().max())
    assert torch.equal(db.cpu(), dy.sum(dim=(0, 1, 2, 3)))
    # input gradient of the plain single-source form (tap-flipped transposed filters)
    if not up0 and C1 == 0:
        wd = torch.empty((27, C0, Cout), dtype=bf, device="cuda")
        wf = torch.empty((27, Cout, C0), dtype=bf, device="cuda")
        ops.pack_weights(w.cuda(), wf, wd)
        dx = torch.empty((N, D, H, W, C0), dtype=bf, device="cuda")
        ops.conv3d_dgrad(dy.to(bf).cuda(), wd, dx)
        assert torch.equal(dx.cpu().view(torch.int16), to_ndhwc(xin.grad).to(bf).view(torch.int16)), "input gradient"


def _upcat_fuzz_cases(n=8, seed=77):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        N, D, H, W = int(rs.choice([1, 2, 3])), int(rs.choice([8, 16, 24])), int(rs.choice([16, 32, 48])), int(rs.choice([32, 64]))
        if N * D * H * W > 80000:
            continue
        out.append((N, D, H, W, int(rs.choice([64, 128])), int(rs.choice([0, 32, 64])), int(rs.choice([64, 128]))))
    return out


@pytest.mark.parametrize("case", _upcat_fuzz_cases(), ids=lambda c: "N%d_%dx%dx%d_up%d+%d_o%d" % c)
def test_parity_form_fuzz_is_exact_on_dyadic_data(ops, case):
    """the parity form of UpSampling3D -> concatenate -> Conv3D on random grids with small dyadic values: the pre-summed filters are exact
    in bf16, every product and partial sum exact in fp32, so both input gradients and the weight gradient must equal the plain definition
    bit for bit, and the forward result the plain definition with the two documented intermediate roundings (with a skip source both
    launches' partial sums pass through bf16 before they meet in fp32)"""
    N, D, H, W, C0, C1, Cout = case
    bf = torch.bfloat16
    if not (ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, bf) & 1):
        pytest.skip("shape outside the parity-form kernels")
    g = torch.Generator().manual_seed(sum(case))
    dy4 = lambda shape, lo, hi, div: (torch.randint(lo, hi, shape, generator=g).float() / div)
    x_low = dy4((N, D // 2, H // 2, W // 2, C0), -4, 5, 4.0)
    x_skip = dy4((N, D, H, W, C1), -4, 5, 4.0) if C1 else None
    w = dy4((27, Cout, C0 + C1), -2, 3, 8.0)
    bias = dy4((Cout,), -4, 5, 4.0)
    up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=bf)
    up_d = torch.empty((8, 8, C0, Cout), device="cuda", dtype=bf)
    sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=bf) if C1 else None
    sk_d = torch.empty((27, C1, Cout), device="cuda", dtype=bf) if C1 else None
    ops.conv3d_pack_up_weights(w.cuda(), C0, C1, up_f, up_d, sk_f, sk_d)
    xl = x_low.clone().requires_grad_(True)
    xs = x_skip.clone().requires_grad_(True) if C1 else None
    wk = keras_kernel_from_packed(w).float().requires_grad_(True)
    pre = F.conv3d(ref_concat_input(xl, xs, True), wk, bias, padding=1)
    y = torch.empty((N, D, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_upcat_fwd(x_low.to(bf).cuda(), None if not C1 else x_skip.to(bf).cuda(), up_f, sk_f, bias.cuda(), y, act=1)
    if C1:
        # two launches: the up-sampled channels' partial sum is stored as bf16, the skip launch rounds its own sum + bias to bf16 and adds the two in fp32
        part = F.conv3d(ref_concat_input(x_low, None, True), wk.detach()[:, :C0].contiguous(), None, padding=1).to(bf).float()
        want = F.relu(part + F.conv3d(to_ncdhw(x_skip), wk.detach()[:, C0:].contiguous(), bias, padding=1).to(bf).float())
    else:
        want = F.relu(pre).detach()
    assert torch.equal(y.cpu().view(torch.int16), to_ndhwc(want).to(bf).view(torch.int16)), "forward"
    dy = dy4((N, D, H, W, Cout), -2, 3, 2.0)
    pre.backward(to_ncdhw(dy))
    dx_low = torch.empty((N, D // 2, H // 2, W // 2, C0), dtype=bf, device="cuda")
    dx_skip = torch.empty((N, D, H, W, C1), dtype=bf, device="cuda") if C1 else None
    ops.conv3d_upcat_dgrad(dy.to(bf).cuda(), up_d, sk_d, None, None, dx_low, dx_skip)
    assert torch.equal(dx_low.cpu().view(torch.int16), xl.grad.to(bf).view(torch.int16)), "gradient of the low-resolution source"
    if C1:
        assert torch.equal(dx_skip.cpu().view(torch.int16), xs.grad.to(bf).view(torch.int16)), "gradient of the skip source"
    if ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, bf) & 2:
        dw = torch.zeros((27, Cout, C0 + C1), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        ops.conv3d_upcat_wgrad(x_low.to(bf).cuda(), None if not C1 else x_skip.to(bf).cuda(), dy.to(bf).cuda(), dw, db, torch.empty(64 * Cout * C0, device="cuda"))
        ref_dw = wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
        assert torch.equal(dw.cpu(), ref_dw), "weight gradient: max diff %g" % float((dw.cpu() - ref_dw).abs().max())
        assert torch.equal(db.cpu(), dy.sum(dim=(0, 1, 2, 3)))


def _planar_fuzz_cases(n=8, seed=5):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        S, H, W = int(rs.choice([1, 3, 8, 13])), int(rs.choice([16, 32, 48, 64])), int(rs.choice([16, 32, 64]))
        if S * H * W > 30000:
            continue
        out.append((S, H, W, int(rs.choice([32, 64, 96])), int(rs.choice([0, 0, 32])), int(rs.choice([32, 64, 128])), bool(rs.rand() < 0.3)))
    return out


@pytest.mark.parametrize("case", _planar_fuzz_cases(), ids=lambda c: "S%d_%dx%d_c%d+%d_o%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_up" if c[6] else ""))
def test_planar_conv_dispatch_fuzz_is_exact_on_dyadic_data(ops, case):
    """2-D slices ride the kernels' D axis (planar = 1: no coupling between slices, 9 live taps): forward, weight gradient and input
    gradient of random slice stacks, bit for bit against the 2-D definition"""
    S, H, W, C0, C1, Cout, up0 = case
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    dy4 = lambda shape, lo, hi, div: (torch.randint(lo, hi, shape, generator=g).float() / div)
    s0 = (1, S, H // 2, W // 2, C0) if up0 else (1, S, H, W, C0)
    x0 = dy4(s0, -4, 5, 4.0)
    x1 = dy4((1, S, H, W, C1), -4, 5, 4.0) if C1 else None
    w = dy4((27, Cout, C0 + C1), -2, 3, 8.0)
    bias = dy4((Cout,), -4, 5, 4.0)
    xin = ref_concat_input(x0, x1, up0, planar=True).float().requires_grad_(True)
    wk = keras_kernel_from_packed(w).float().requires_grad_(True)
    yref = F.conv3d(xin, planar_kernel(wk), bias, padding=1)
    y = torch.empty((1, S, H, W, Cout), dtype=bf, device="cuda")
    ops.conv3d_fwd(x0.to(bf).cuda(), None if x1 is None else x1.to(bf).cuda(), w.to(bf).cuda(), bias.cuda(), y, up0=up0, act=1, planar=True)
    assert torch.equal(y.cpu().view(torch.int16), to_ndhwc(F.relu(yref).detach()).to(bf).view(torch.int16)), "forward"
    dy = dy4((1, S, H, W, Cout), -2, 3, 2.0)
    yref.backward(to_ncdhw(dy))
    dw = torch.zeros((27, Cout, C0 + C1), dtype=torch.float32, device="cuda")
    db = torch.zeros((Cout,), dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(x0.to(bf).cuda(), None if x1 is None else x1.to(bf).cuda(), dy.to(bf).cuda(), dw, db, up0=up0, planar=True)
    ref_dw = wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1)
    assert torch.equal(dw.cpu()[9:18], ref_dw[9:18]), "weight gradient (centre kd plane)"
    assert float(dw.cpu()[:9].abs().max()) == 0.0 and float(dw.cpu()[18:].abs().max()) == 0.0      # the other planes do not exist in 2-D
    assert torch.equal(db.cpu(), dy.sum(dim=(0, 1, 2, 3)))
    if not up0 and C1 == 0:
        wd = torch.empty((27, C0, Cout), dtype=bf, device="cuda")
        wf = torch.empty((27, Cout, C0), dtype=bf, device="cuda")
        ops.pack_weights(w.cuda(), wf, wd)
        dx = torch.empty((1, S, H, W, C0), dtype=bf, device="cuda")
        ops.conv3d_dgrad(dy.to(bf).cuda(), wd, dx, planar=True)
        assert torch.equal(dx.cpu().view(torch.int16), to_ndhwc(xin.grad).to(bf).view(torch.int16)), "input gradient"
