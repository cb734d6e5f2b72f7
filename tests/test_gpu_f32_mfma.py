"""The fp32 parity mode ON THE BENCHMARKED KERNEL STRUCTURE (round 6): fp32 tensors and filters on the fp32 instantiation
of the warp-specialised MFMA kernels (k_conv_fwd_ws<..., F32> on v_mfma_f32_32x32x2_f32, k_conv_wgrad_kd<.., F32>): same halo box, LDS-DMA
pieces, swizzle, phases, asynchronous drain and tails as the bf16 launches, 16-channel chunks.  An fp32 MFMA is an fmaf chain per output:
against fp64 only the summation order and fp32 rounding remain.

 (i)   every kernel form against torch-CPU fp64: plain / dual-source / fused up-sampling forward, the input-gradient form with the ReLU
       mask, the pooled-copy tail, the parity form (forward, both input gradients), the weight gradient;
 (ii)  `impl=IMPL_MFMA` makes the call FAIL where the MFMA path does not take it: a silent fall-back to the VALU kernels cannot pass;
 (iii) the north-star bar at BASELINE configs[1] size: logits <= 1e-3 relative, Dice <= 1e-4 against the CPU oracle (reference
       fetal_net/model/unet3d/unet.py:68, metrics.py:11-15) - tests/test_gpu_fullsize_parity.py::test_n1_full_size_fp32_on_the_mfma_kernels_vs_oracle.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_close, f64, keras_kernel_from_packed, ref_concat_input, ref_conv_fwd, rnd, to_ncdhw, to_ndhwc

pytestmark = pytest.mark.gpu

F32 = torch.float32
TOL = (1e-5, 1e-5)       # |got - ref| <= 1e-5 max|ref| + 1e-5 |ref|: fp32 chains of up to 27 x 128 products (measured: see the printed maxima)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from fmri_hip import ops as o
    return o


CASES = [
    # name, N, D, H, W, C0, up0, C1, Cout
    ("one_chunk_16_32", 1, 4, 8, 16, 16, False, 0, 32),
    ("two_chunks_multi_tile", 2, 8, 16, 32, 32, False, 0, 64),
    ("eight_chunks_border_tile", 1, 4, 8, 16, 128, False, 0, 32),
    ("dual_source", 1, 4, 8, 16, 16, False, 32, 64),
    ("dual_source_fused_upsampling", 1, 8, 16, 32, 32, True, 16, 64),
    ("many_tiles_per_workgroup", 4, 16, 32, 64, 16, False, 0, 32),          # 1,024 (tile, block) pairs on 256 workgroups: the asynchronous drain under the next tile
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_bias_relu(ops, case):
    name, N, D, H, W, C0, up0, C1, Cout = case
    from fmri_hip._lib import IMPL_MFMA, lib
    assert lib().fmri_conv3d_uses_mfma(C0, C1, Cout, D, H, W, 0) & 1
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = rnd(s0, 1, F32)
    src1 = rnd((N, D, H, W, C1), 2, F32) if C1 else None
    w = rnd((27, Cout, C0 + C1), 3, F32, scale=0.2)
    bias = rnd((Cout,), 4, F32)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=F32, device="cuda")
    ops.conv3d_fwd(src0, src1, w, bias, y, up0=up0, act=1, impl=IMPL_MFMA)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(src0), None if src1 is None else f64(src1), up0, f64(w), f64(bias), 1)
    assert_close(y, ref, *TOL, what=name)
    # and the VALU kernels of the fp32 mode of rounds 1-5 give the same tensor up to summation order
    y2 = torch.empty_like(y)
    ops.conv3d_fwd(src0, src1, w, bias, y2, up0=up0, act=1, impl=1)
    assert_close(y, f64(y2), *TOL, what=name + " vs generic kernel")


@pytest.mark.parametrize("act,alpha", [(0, 0.0), (2, 0.01)], ids=["no_activation", "leaky_relu"])
def test_forward_three_chunks_three_blocks_other_activations(ops, act, alpha):
    """48 input channels = three 16-channel chunks (the second halo slot is re-used within a tile), 96 output channels = three 32-wide blocks,
    three samples; the epilogue's other two activations (none: every input-gradient launch; LeakyReLU: the Isensee models)"""
    from fmri_hip._lib import IMPL_MFMA
    N, D, H, W, C0, Cout = 3, 8, 16, 32, 48, 96
    x = rnd((N, D, H, W, C0), 31, F32)
    w = rnd((27, Cout, C0), 32, F32, scale=0.2)
    bias = rnd((Cout,), 33, F32)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=F32, device="cuda")
    ops.conv3d_fwd(x, None, w, bias, y, act=act, alpha=alpha, impl=IMPL_MFMA)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(x), None, False, f64(w), f64(bias), 0)
    if act == 2:
        ref = torch.where(ref > 0, ref, alpha * ref)
    assert_close(y, ref, *TOL, what="act %d" % act)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_input_gradient_form_with_relu_mask(ops, case):
    name, N, D, H, W, C0, up0, C1, Cout = case
    from fmri_hip._lib import IMPL_MFMA
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = rnd(s0, 5, F32)
    src1 = rnd((N, D, H, W, C1), 6, F32) if C1 else None
    w = rnd((27, Cout, C0 + C1), 7, F32, scale=0.2)
    mask = rnd((N, D, H, W, Cout), 8, F32)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=F32, device="cuda")
    ops.conv3d_fwd(src0, src1, w, None, y, up0=up0, act=0, mask=mask, impl=IMPL_MFMA)
    torch.cuda.synchronize()
    ref = ref_conv_fwd(f64(src0), None if src1 is None else f64(src1), up0, f64(w), None, 0) * (f64(mask) > 0)
    assert_close(y, ref, *TOL, what=name)


@pytest.mark.parametrize("shape", [(1, 8, 16, 32, 16, 32), (2, 16, 32, 64, 32, 64)], ids=lambda c: "N%d_%dx%dx%d_%d_%d" % c)
def test_pooled_copy_tail(ops, shape):
    """fmri_conv3d_fwd_tail in fp32: the drain also writes MaxPooling3D(2)(y) (reference unet3d/unet.py:45-51) - y bit-identical to the plain
    launch, the pooled tensor bit-identical to fmri_maxpool3d_2x_fwd(y); no logits tail in fp32 (32-wide blocks)"""
    N, D, H, W, C0, Cout = shape
    assert ops.conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, F32) == 1
    x = rnd((N, D, H, W, C0), 21, F32)
    w = rnd((27, Cout, C0), 22, F32, scale=0.1)
    bias = rnd((Cout,), 23, F32, scale=0.3)
    y0 = torch.empty((N, D, H, W, Cout), dtype=F32, device="cuda")
    ops.conv3d_fwd(x, None, w, bias, y0, act=1)
    p0 = torch.empty((N, D // 2, H // 2, W // 2, Cout), dtype=F32, device="cuda")
    ops.maxpool_fwd(y0, p0)
    y = torch.full_like(y0, float("nan"))
    pool = torch.full_like(p0, float("nan"))
    ops.conv3d_fwd_tail(x, w, bias, y, pool=pool, act=1)
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and torch.equal(pool, p0)
    assert_close(y, ref_conv_fwd(f64(x), None, False, f64(w), f64(bias), 1), *TOL, what="tail y")
    with pytest.raises(RuntimeError):
        ops.conv3d_fwd_tail(x, w, bias, y, w1=bias, b1=bias[:1], logits=torch.empty(N * D * H * W, dtype=F32, device="cuda"))


# name, N, D, H, W (output dims), C0 (up-sampled), C1 (skip), Cout: every channel count is also an OUTPUT width of one of the launches (the input
# gradients), i.e. a multiple of the 32-wide block
UPCAT = [("one_tile", 1, 8, 16, 32, 32, 32, 32), ("multi_tile", 2, 16, 32, 64, 64, 32, 64), ("deep", 1, 8, 16, 32, 128, 64, 64)]


@pytest.mark.parametrize("case", UPCAT, ids=[c[0] for c in UPCAT])
def test_parity_form_forward_and_input_gradients(ops, case):
    """fmri_conv3d_upcat_fwd / _dgrad in fp32 (MODE 1 / MODE 2 launches on the tight box + the residual drain, reference unet.py:132-138,61,102)
    against the plain definition in fp64.  The pre-summed filters are fp32 sums of up to 8 fp32 taps: one more rounding at 6e-8."""
    name, N, D, H, W, C0, C1, Cout = case
    assert ops.conv3d_upcat_ok(C0, C1, Cout, D, H, W, F32) & 1
    x_low = rnd((N, D // 2, H // 2, W // 2, C0), 1, F32)
    x_skip = rnd((N, D, H, W, C1), 2, F32)
    w = rnd((27, Cout, C0 + C1), 3, F32, scale=0.05)
    bias = rnd((Cout,), 4, F32)
    up_f = torch.empty((8, 8, Cout, C0), device="cuda", dtype=F32)
    up_d = torch.empty((8, 8, C0, Cout), device="cuda", dtype=F32)
    sk_f = torch.empty((27, Cout, C1), device="cuda", dtype=F32)
    sk_d = torch.empty((27, C1, Cout), device="cuda", dtype=F32)
    ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=F32, device="cuda")
    ops.conv3d_upcat_fwd(x_low, x_skip, up_f, sk_f, bias, y, act=1)
    torch.cuda.synchronize()
    xl = f64(x_low).requires_grad_(True)
    xs = f64(x_skip).requires_grad_(True)
    pre = F.conv3d(ref_concat_input(xl, xs, True), keras_kernel_from_packed(f64(w)), f64(bias), padding=1)
    assert_close(y, to_ndhwc(F.relu(pre)), *TOL, what=name + " fwd")
    dy = rnd((N, D, H, W, Cout), 5, F32)
    m_low = rnd((N, D // 2, H // 2, W // 2, C0), 6, F32).clamp_min(0)
    m_skip = rnd((N, D, H, W, C1), 7, F32).clamp_min(0)
    dx_low = torch.full_like(x_low, float("nan"))
    dx_skip = torch.full_like(x_skip, float("nan"))
    ops.conv3d_upcat_dgrad(dy, up_d, sk_d, m_low, m_skip, dx_low, dx_skip)
    torch.cuda.synchronize()
    pre.backward(to_ncdhw(f64(dy)))
    assert_close(dx_low, xl.grad * (f64(m_low) > 0), *TOL, what=name + " dx_low")
    assert_close(dx_skip, xs.grad * (f64(m_skip) > 0), *TOL, what=name + " dx_skip")


# (32 x 32 blocks: every source's channel count and Cout are multiples of 32)
WG = [("one_block", 2, 8, 16, 32, 32, 0, 32), ("two_by_two_blocks", 1, 8, 32, 32, 64, 0, 64), ("dual_source", 1, 4, 16, 32, 32, 32, 32),
      ("column_crossing", 5, 4, 32, 32, 32, 0, 32), ("fused_upsampling_dual", 1, 8, 16, 32, 64, 32, 32)]


@pytest.mark.parametrize("case", WG, ids=[c[0] for c in WG])
def test_weight_gradient(ops, case):
    name, N, D, H, W, C0, C1, Cout = case
    from fmri_hip._lib import IMPL_MFMA, lib
    assert lib().fmri_conv3d_uses_mfma(C0, C1, Cout, D, H, W, 0) & 2
    up0 = name.startswith("fused_upsampling")
    src0 = rnd((N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0), 9, F32)
    src1 = rnd((N, D, H, W, C1), 10, F32) if C1 else None
    dy = rnd((N, D, H, W, Cout), 11, F32)
    dw = torch.zeros((27, Cout, C0 + C1), dtype=F32, device="cuda")
    db = torch.zeros((Cout,), dtype=F32, device="cuda")
    ops.conv3d_wgrad(src0, src1, dy, dw, db, up0=up0, impl=IMPL_MFMA)
    torch.cuda.synchronize()
    x = ref_concat_input(f64(src0), None if src1 is None else f64(src1), up0)
    wk = torch.zeros((Cout, C0 + C1, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv3d(x, wk, None, padding=1).backward(to_ncdhw(f64(dy)))
    assert_close(dw, wk.grad.permute(2, 3, 4, 0, 1).reshape(27, Cout, C0 + C1), 2e-5, 2e-5, what=name + " dw")
    assert_close(db, f64(dy).sum(dim=(0, 1, 2, 3)), 2e-5, 2e-5, what=name + " db")
