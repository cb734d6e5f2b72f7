"""torch-tensor front-ends of the C ABI.  torch is plumbing only: device memory, the current HIP stream, dtypes."""
import torch

from ._lib import lib, check, F32, BF16, ACT_NONE, ACT_RELU, IMPL_AUTO


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


import os as _os

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) if _os.environ.get("FMRI_RAW_STREAM", "1") != "0" else None
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    """the current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() costs 10-40 us of Python per call
    (device-index resolution, an is_available() probe with an environment lookup, a Stream object); an engine step makes ~100 of these
    calls and the patch sampler is launch-bound, so the two C entry points torch itself uses on its fast paths are called directly."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fmri_hip ops need device tensors (no CPU path)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("fmri_hip ops need contiguous tensors")


def conv3d_fwd(src0, src1, w, bias, y, up0=False, act=ACT_RELU, alpha=0.0, mask=None, impl=IMPL_AUTO, planar=False):
    """src0 [N,d,h,w,C0] (half-res when up0), src1 [N,D,H,W,C1] or None, w [27,Cout,C0+C1], y [N,D,H,W,Cout]."""
    _need_cuda(src0, src1, w, bias, y, mask)
    N, D, H, W, Cout = y.shape
    C0 = src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    assert w.shape == (27, Cout, C0 + C1), (w.shape, Cout, C0, C1)
    check(lib().fmri_conv3d_fwd(_p(src0), C0, int(up0), _p(src1), C1, _p(w), _p(bias), _p(mask), _p(y), N, D, H, W, Cout,
                                act, float(alpha), dt(y), impl, int(planar), _s()), "fmri_conv3d_fwd")
    return y


def conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, dtype, planar=False):
    """bit 0: the conv epilogue can also write the 2x2x2 (planar: 2x2 per slice) max-pooled tensor, bit 1: ... the logits of a final
    1x1x1 conv to one label"""
    f = lib().fmri_conv3d_fwd_tail_planar_ok if planar else lib().fmri_conv3d_fwd_tail_ok
    return int(f(C0, Cout, N, D, H, W, BF16 if dtype == torch.bfloat16 else F32))


def conv3d_fwd_tail(src0, w, bias, y, pool=None, w1=None, b1=None, logits=None, act=ACT_RELU, alpha=0.0, planar=False):
    """conv block whose epilogue also produces MaxPooling3D(2)(y) (planar: MaxPooling2D(2) of every slice) and / or the final 1x1x1 conv's
    logits (fmri_conv3d_fwd_tail / fmri_conv3d_fwd_tail_planar)"""
    _need_cuda(src0, w, bias, y, pool, w1, b1, logits)
    N, D, H, W, Cout = y.shape
    C0 = src0.shape[-1]
    assert w.shape == (27, Cout, C0)
    if pool is not None:
        assert tuple(pool.shape) == (N, D if planar else D // 2, H // 2, W // 2, Cout) and pool.dtype == y.dtype
    if logits is not None:
        assert logits.dtype == torch.float32 and logits.numel() == N * D * H * W and w1.dtype == torch.float32 and w1.numel() == Cout
    f = lib().fmri_conv3d_fwd_tail_planar if planar else lib().fmri_conv3d_fwd_tail
    check(f(_p(src0), C0, _p(w), _p(bias), _p(y), _p(pool), _p(w1), _p(b1), _p(logits), N, D, H, W, Cout, act, float(alpha), dt(y), _s()),
          "fmri_conv3d_fwd_tail_planar" if planar else "fmri_conv3d_fwd_tail")
    return y


def conv3d_dgrad(dy, w_dgrad, dx, mask=None, impl=IMPL_AUTO, planar=False):
    _need_cuda(dy, w_dgrad, dx, mask)
    N, D, H, W, Cin = dx.shape
    Cout = dy.shape[-1]
    assert w_dgrad.shape == (27, Cin, Cout)
    check(lib().fmri_conv3d_dgrad(_p(dy), Cout, _p(w_dgrad), _p(mask), _p(dx), N, D, H, W, Cin, dt(dx), impl, int(planar), _s()),
          "fmri_conv3d_dgrad")
    return dx


def conv3d_wgrad(src0, src1, dy, dw, db, up0=False, impl=IMPL_AUTO, planar=False, workspace=None):
    _need_cuda(src0, src1, dy, dw, db)
    N, D, H, W, Cout = dy.shape
    C0 = src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    assert dw.dtype == torch.float32 and dw.numel() == 27 * Cout * (C0 + C1)
    check(lib().fmri_conv3d_wgrad(_p(src0), C0, int(up0), _p(src1), C1, _p(dy), _p(dw), _p(db), N, D, H, W, Cout, dt(dy),
                                  impl, int(planar), _p(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(),
                                  _s()), "fmri_conv3d_wgrad")


def pack_weights(w, w_fwd, w_dgrad):
    _need_cuda(w, w_fwd, w_dgrad)
    _, Cout, Cin = w.shape
    d = dt(w_fwd if w_fwd is not None else w_dgrad)
    check(lib().fmri_conv3d_pack_weights(_p(w), _p(w_fwd), _p(w_dgrad), Cout, Cin, d, _s()), "fmri_conv3d_pack_weights")


def pack_table(entries, device):
    """the device table of fmri_pack_weights_batched.  entries: ("plain", w, w_fwd, w_dgrad) or ("up", w, C0, C1, up_f, up_d, sk_f, sk_d, planar)
    with the tensors of pack_weights / conv3d_pack_up_weights (None = not wanted).  Returns (table int64 [n][10], n_blocks); the table holds
    raw device addresses: the tensors must stay alive and in place."""
    rows, first = [], 0
    ptr = lambda t: 0 if t is None else t.data_ptr()
    for e in entries:
        if e[0] == "plain":
            _, w, wf, wd = e
            _, Cout, Cin = w.shape
            rows.append([0, first, ptr(w), ptr(wf), ptr(wd), 0, 0, Cout, Cin, 0])
            first += 27 * ((Cout + 63) // 64) * ((Cin + 63) // 64)
        else:
            _, w, C0, C1, up_f, up_d, sk_f, sk_d, planar = e
            Cout = w.shape[1]
            rows.append([1, first, ptr(w), ptr(up_f), ptr(up_d), ptr(sk_f), ptr(sk_d), Cout, C0, C1 | (int(bool(planar)) << 32)])
            tco = (Cout + 63) // 64
            first += (16 if planar else 64) * tco * ((C0 + 63) // 64) + 27 * tco * ((C1 + 63) // 64)
    return torch.tensor(rows, dtype=torch.int64, device=device), first


def pack_weights_batched(table, n_blocks, dtype):
    """every weight image of a model in one launch (fmri_pack_weights_batched); table, n_blocks from pack_table"""
    _need_cuda(table)
    check(lib().fmri_pack_weights_batched(_p(table), table.shape[0], n_blocks, BF16 if dtype == torch.bfloat16 else F32, _s()),
          "fmri_pack_weights_batched")


def conv1x1_fwd(x, w, b, logits):
    _need_cuda(x, w, b, logits)
    C = x.shape[-1]
    L = w.shape[0]
    nvox = x.numel() // C
    check(lib().fmri_conv1x1_fwd(_p(x), _p(w), _p(b), _p(logits), nvox, C, L, dt(x), _s()), "fmri_conv1x1_fwd")
    return logits


def conv1x1_bwd(x, w, dlogits, dx, dw, db, relu_mask=True):
    _need_cuda(x, w, dlogits, dx, dw, db)
    C = x.shape[-1]
    L = w.shape[0]
    nvox = x.numel() // C
    check(lib().fmri_conv1x1_bwd(_p(x), _p(w), _p(dlogits), _p(dx), _p(dw), _p(db), nvox, C, L, int(relu_mask), dt(x), _s()),
          "fmri_conv1x1_bwd")


def sigmoid_dice_fwd(logits, y_true, probs, sums, weight=None):
    """weight (optional, fp32, one per voxel): multiplies the cross-entropy term (dice_and_xent_mask)"""
    _need_cuda(logits, y_true, probs, sums, weight)
    assert y_true.dtype == torch.uint8 and sums.dtype == torch.float64 and sums.numel() >= 16
    if weight is not None:
        assert weight.dtype == torch.float32 and weight.numel() == logits.numel()
        check(lib().fmri_sigmoid_dice_fwd_weighted(_p(logits), _p(y_true), _p(weight), _p(probs), _p(sums), logits.numel(), _s()),
              "fmri_sigmoid_dice_fwd_weighted")
        return
    check(lib().fmri_sigmoid_dice_fwd(_p(logits), _p(y_true), _p(probs), _p(sums), logits.numel(), _s()), "fmri_sigmoid_dice_fwd")


def sigmoid_dice_bwd(probs, y_true, sums, dlogits, smooth=1.0, grad_scale=1.0):
    _need_cuda(probs, y_true, sums, dlogits)
    check(lib().fmri_sigmoid_dice_bwd(_p(probs), _p(y_true), _p(sums), _p(dlogits), probs.numel(), float(smooth),
                                      float(grad_scale), _s()), "fmri_sigmoid_dice_bwd")


def maxpool_fwd(x, y, planar=False):
    _need_cuda(x, y)
    N, D, H, W, Cc = x.shape
    check(lib().fmri_maxpool3d_2x_fwd(_p(x), _p(y), N, D, H, W, Cc, dt(x), int(planar), _s()), "fmri_maxpool3d_2x_fwd")
    return y


def maxpool_bwd(x, dy, dx, add=None, add_off=0, relu_mask=True, planar=False):
    _need_cuda(x, dy, dx, add)
    N, D, H, W, Cc = x.shape
    add_ld = 0 if add is None else add.shape[-1]
    check(lib().fmri_maxpool3d_2x_bwd(_p(x), _p(dy), _p(add), add_ld, add_off, _p(dx), N, D, H, W, Cc, int(relu_mask), dt(x), int(planar), _s()),
          "fmri_maxpool3d_2x_bwd")
    return dx


def upsample_fwd(x, y, y_off=0, planar=False):
    _need_cuda(x, y)
    N, D, H, W, Cc = x.shape
    check(lib().fmri_upsample_nearest2x_fwd(_p(x), _p(y), y.shape[-1], y_off, N, D, H, W, Cc, dt(x), int(planar), _s()),
          "fmri_upsample_nearest2x_fwd")
    return y


def upsample_bwd(dy, dx, dy_off=0, xmask=None, planar=False):
    _need_cuda(dy, dx, xmask)
    N, D, H, W, Cc = dx.shape
    check(lib().fmri_upsample_nearest2x_bwd(_p(dy), dy.shape[-1], dy_off, _p(xmask), _p(dx), N, D, H, W, Cc, dt(dx), int(planar), _s()),
          "fmri_upsample_nearest2x_bwd")
    return dx


def adam_step(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    _need_cuda(p, g, m, v)
    check(lib().fmri_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr_t), beta1, beta2, eps, float(grad_scale), _s()),
          "fmri_adam_step")


def tile_gather(vol, idx, patch, tiles):
    _need_cuda(vol, idx, tiles)
    X, Y, Z = vol.shape
    B = idx.shape[0]
    check(lib().fmri_tile_gather(_p(vol), X, Y, Z, _p(idx), B, patch[0], patch[1], patch[2], _p(tiles), dt(tiles), _s()),
          "fmri_tile_gather")
    return tiles


def tile_scatter_accumulate(pred, idx, patch, acc, cnt):
    _need_cuda(pred, idx, acc, cnt)
    X, Y, Z, Cc = acc.shape
    B = idx.shape[0]
    check(lib().fmri_tile_scatter_accumulate(_p(pred), _p(idx), B, patch[0], patch[1], patch[2], Cc, _p(acc), _p(cnt), X, Y, Z,
                                             _s()), "fmri_tile_scatter_accumulate")


def tile_finalize(acc, cnt, out, bad):
    _need_cuda(acc, cnt, out, bad)
    Cc = acc.shape[-1]
    check(lib().fmri_tile_finalize(_p(acc), _p(cnt), _p(out), _p(bad), cnt.numel(), Cc, _s()), "fmri_tile_finalize")


def cast(src, dst):
    _need_cuda(src, dst)
    check(lib().fmri_cast(_p(src), dt(src), _p(dst), dt(dst), src.numel(), _s()), "fmri_cast")
    return dst


def norm_act_fwd(x, gamma, beta, y, stats, ws, per_instance, eps=1e-3, eps_on_std=False, act=ACT_RELU, alpha=0.0):
    """x,y [N,...,C]; stats [G,C,3] fp32; ws [G,C,2] fp64 scratch"""
    _need_cuda(x, gamma, beta, y, stats, ws)
    N, Cc = x.shape[0], x.shape[-1]
    V = x.numel() // (N * Cc)
    check(lib().fmri_norm_act_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), N, V, Cc, int(per_instance), float(eps),
                                  int(eps_on_std), act, float(alpha), dt(x), _s()), "fmri_norm_act_fwd")
    return y


def norm_act_bwd(x, y, dy, gamma, stats, dx, dgamma, dbeta, ws, per_instance, act=ACT_RELU, alpha=0.0, beta=None):
    """beta given: the sign of the block's output comes from x (recomputed exactly as the forward formed it), y is not read (may be None)"""
    _need_cuda(x, y, dy, gamma, stats, dx, dgamma, dbeta, ws, beta)
    N, Cc = x.shape[0], x.shape[-1]
    V = x.numel() // (N * Cc)
    if beta is not None:
        check(lib().fmri_norm_act_bwd_x(_p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(dx), _p(dgamma), _p(dbeta), _p(ws), N, V, Cc,
                                        int(per_instance), act, float(alpha), dt(x), _s()), "fmri_norm_act_bwd_x")
        return dx
    check(lib().fmri_norm_act_bwd(_p(x), _p(y), _p(dy), _p(gamma), _p(stats), _p(dx), _p(dgamma), _p(dbeta), _p(ws), N, V, Cc,
                                  int(per_instance), act, float(alpha), dt(x), _s()), "fmri_norm_act_bwd")
    return dx


# ---- normalisation tails of the conv launches (include/fmri_hip.h: "Normalisation tails")
def conv3d_fwd_ntail_ok(C0, C1, Cout, N, D, H, W, dtype):
    return bool(lib().fmri_conv3d_fwd_ntail_ok(C0, C1, Cout, N, D, H, W, BF16 if dtype == torch.bfloat16 else F32))


def norm_tail_ws_doubles(G, C):
    """fp64 elements of the `ws` the tail entry points need (the totals [G,C,2] in front + one block per workgroup of the persistent launch)"""
    return int(lib().fmri_norm_tail_ws_doubles(int(G), int(C)))


def conv3d_fwd_stats(src0, src1, w, bias, y, ws, per_instance, up0=False, act=ACT_NONE, alpha=0.0):
    """conv3d_fwd whose epilogue leaves {sum y, sum y^2} per (group, channel) in front of ws (fp64, norm_tail_ws_doubles(G, Cout) elements).
    ws must be ZERO on entry (allocate it zeroed): no fmri_norm_* / *_stats call clears it in front any more - the last reader of the sums
    (norm_act_fwd_pre / norm_act_bwd_pre / the fold kernel) leaves it zero for the next call.  After a call sequence that was interrupted
    between the summing launch and its reader, zero it again before reuse."""
    _need_cuda(src0, src1, w, bias, y, ws)
    N, D, H, W, Cout = y.shape
    c0, c1 = src0.shape[-1], (0 if src1 is None else src1.shape[-1])
    check(lib().fmri_conv3d_fwd_stats(_p(src0), c0, int(up0), _p(src1), c1, _p(w), _p(bias), _p(y), N, D, H, W, Cout, act, float(alpha), _p(ws),
                                      int(per_instance), dt(y), _s()), "fmri_conv3d_fwd_stats")
    return y


def conv3d_upcat_fwd_stats(src0_low, src1, w_up_f, w_sk_f, bias, y, ws, per_instance, act=ACT_NONE, alpha=0.0):
    _need_cuda(src0_low, src1, w_up_f, w_sk_f, bias, y, ws)
    N, D, H, W, Cout = y.shape
    check(lib().fmri_conv3d_upcat_fwd_stats(_p(src0_low), src0_low.shape[-1], _p(src1), src1.shape[-1], _p(w_up_f), _p(w_sk_f), _p(bias), _p(y),
                                            N, D, H, W, Cout, act, float(alpha), _p(ws), int(per_instance), dt(y), _s()),
          "fmri_conv3d_upcat_fwd_stats")
    return y


def norm_act_fwd_pre(x, gamma, beta, y, stats, ws, per_instance, eps=1e-3, eps_on_std=False, act=ACT_RELU, alpha=0.0):
    """norm_act_fwd with the sums {sum x, sum x^2} already in ws (conv3d_fwd_stats / conv3d_upcat_fwd_stats)"""
    _need_cuda(x, gamma, beta, y, stats, ws)
    N, Cc = x.shape[0], x.shape[-1]
    V = x.numel() // (N * Cc)
    check(lib().fmri_norm_act_fwd_pre(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), N, V, Cc, int(per_instance), float(eps),
                                      int(eps_on_std), act, float(alpha), dt(x), _s()), "fmri_norm_act_fwd_pre")
    return y


def norm_moving_update(stats, moving_mean, moving_var, M, momentum=0.99, eps=1e-3):
    """Keras BatchNormalization moving mean / variance after a training forward (one launch)"""
    _need_cuda(stats, moving_mean, moving_var)
    check(lib().fmri_norm_moving_update(_p(stats), _p(moving_mean), _p(moving_var), moving_mean.numel(), float(M), float(momentum), float(eps), _s()),
          "fmri_norm_moving_update")


def norm_scale_shift(stats, gamma, beta, nss):
    """nss [G,C,2] fp32 = {scale, shift} of the apply pass (z = fma(x, scale, shift))"""
    _need_cuda(stats, gamma, beta, nss)
    check(lib().fmri_norm_scale_shift(_p(stats), _p(gamma), _p(beta), _p(nss), stats.shape[0], stats.shape[1], _s()), "fmri_norm_scale_shift")
    return nss


def conv3d_dgrad_norm(dy, w_dgrad, x, nss, dz, ws, per_instance, act=ACT_RELU, alpha=0.0):
    """conv3d_dgrad into a normalised block: dz = dgrad * act'(z(x)), ws [G,Cin,2] = {sum dz, sum dz * x}"""
    _need_cuda(dy, w_dgrad, x, nss, dz, ws)
    N, D, H, W, Cin = dz.shape
    check(lib().fmri_conv3d_dgrad_norm(_p(dy), dy.shape[-1], _p(w_dgrad), _p(x), _p(nss), _p(dz), N, D, H, W, Cin, act, float(alpha), _p(ws),
                                       int(per_instance), dt(dz), _s()), "fmri_conv3d_dgrad_norm")
    return dz


def norm_act_bwd_pre(x, dz, gamma, stats, dx, dgamma, dbeta, ws, per_instance):
    _need_cuda(x, dz, gamma, stats, dx, dgamma, dbeta, ws)
    N, Cc = x.shape[0], x.shape[-1]
    V = x.numel() // (N * Cc)
    check(lib().fmri_norm_act_bwd_pre(_p(x), _p(dz), _p(gamma), _p(stats), _p(dx), _p(dgamma), _p(dbeta), _p(ws), N, V, Cc, int(per_instance),
                                      dt(x), _s()), "fmri_norm_act_bwd_pre")
    return dx


def deconv_fwd(x, w, b, y, planar=False):
    """x [N,D,H,W,Cin], w [8,Cout,Cin] (compute dtype), y [N,2D,2H,2W,Cout]"""
    _need_cuda(x, w, b, y)
    N, D, H, W, Cin = x.shape
    Cout = w.shape[1]
    check(lib().fmri_deconv3d_k2s2_fwd(_p(x), _p(w), _p(b), _p(y), N, D, H, W, Cin, Cout, dt(x), int(planar), _s()),
          "fmri_deconv3d_k2s2_fwd")
    return y


def deconv_bwd(x, w, dy, dx, dw, db, dy_off=0, xmask=None, planar=False):
    _need_cuda(x, w, dy, dx, dw, db, xmask)
    N, D, H, W, Cin = x.shape
    Cout = w.shape[1]
    check(lib().fmri_deconv3d_k2s2_bwd(_p(x), _p(w), _p(dy), dy.shape[-1], dy_off, _p(xmask), _p(dx), _p(dw), _p(db), N, D, H, W, Cin,
                                       Cout, dt(x), int(planar), _s()), "fmri_deconv3d_k2s2_bwd")


def conv_direct_fwd(x, w, bias, y, ksize, stride, act=ACT_NONE, alpha=0.0, planar=False):
    """x [N,D,H,W,Cin], w [k^3,Cout,Cin] (compute dtype), y [N,ceil(D/s),ceil(H/s),ceil(W/s),Cout]; planar: x [1,S,H,W,Cin] 2-D slices,
    the stride applies to H and W only"""
    _need_cuda(x, w, bias, y)
    N, D, H, W, Cin = x.shape
    Cout = w.shape[1]
    if planar:
        assert N == 1
        check(lib().fmri_conv2d_direct_fwd(_p(x), _p(w), _p(bias), _p(y), D, H, W, Cin, Cout, ksize, stride, act, float(alpha), dt(x), _s()),
              "fmri_conv2d_direct_fwd")
        return y
    check(lib().fmri_conv3d_direct_fwd(_p(x), _p(w), _p(bias), _p(y), N, D, H, W, Cin, Cout, ksize, stride, act, float(alpha), dt(x), _s()),
          "fmri_conv3d_direct_fwd")
    return y


def conv_direct_bwd(x, w, dy, dx, dw, db, ksize, stride, planar=False):
    _need_cuda(x, w, dy, dx, dw, db)
    N, D, H, W, Cin = x.shape
    Cout = w.shape[1]
    if planar:
        assert N == 1
        check(lib().fmri_conv2d_direct_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), D, H, W, Cin, Cout, ksize, stride, dt(x), _s()),
              "fmri_conv2d_direct_bwd")
        return
    check(lib().fmri_conv3d_direct_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), N, D, H, W, Cin, Cout, ksize, stride, dt(x), _s()),
          "fmri_conv3d_direct_bwd")


def add(a, b, y):
    _need_cuda(a, b, y)
    check(lib().fmri_add(_p(a), _p(b), _p(y), a.numel(), dt(a), _s()), "fmri_add")
    return y


def channel_scale(x, scale, y):
    """x,y [N,...,C]; scale fp32 [N,C]"""
    _need_cuda(x, scale, y)
    N, Cc = x.shape[0], x.shape[-1]
    check(lib().fmri_channel_scale(_p(x), _p(scale), _p(y), N, x.numel() // (N * Cc), Cc, dt(x), _s()), "fmri_channel_scale")
    return y


def slice_channels(src, off, dst, accumulate=False):
    """dst [...,C] (+)= src[..., off:off+C]"""
    _need_cuda(src, dst)
    Cc = dst.shape[-1]
    check(lib().fmri_slice_channels(_p(src), src.shape[-1], off, _p(dst), Cc, dst.numel() // Cc, int(accumulate), dt(dst), _s()),
          "fmri_slice_channels")
    return dst


def act_bwd(y, dy, dx, act, alpha=0.0):
    _need_cuda(y, dy, dx)
    check(lib().fmri_act_bwd(_p(y), _p(dy), _p(dx), act, float(alpha), y.numel(), dt(y), _s()), "fmri_act_bwd")
    return dx


LOSS_KINDS = {"dice_coefficient_loss": 0, "binary_crossentropy_loss": 1, "dice_and_xent": 2, "focal_loss": 3, "vod_coefficient_loss": 4,
              "double_dice_loss": 5, "weighted_dice_coefficient_loss": 6}
LOSS_WEIGHTED_DICE = 6          # not a kind of fmri_sigmoid_loss_bwd: per-(sample, label) sums, fmri_weighted_dice_fwd / _bwd
WEIGHTED_DICE_SMOOTH = 1e-5     # reference metrics.py:39


def weighted_dice_fwd(probs, y_true, gsums, sums, nsamples, n_labels, smooth=WEIGHTED_DICE_SMOOTH):
    """per-(sample, label) Dice sums of probs / y_true [(n * vox + v) * L + l] into gsums [nsamples * L, 3] (float64, zeroed by the call);
    sums[10] += sum of the groups' Dice coefficients, sums[11] += group count (reference metrics.py:39-51)"""
    _need_cuda(probs, y_true, gsums, sums)
    assert y_true.dtype == torch.uint8 and gsums.dtype == torch.float64 and sums.dtype == torch.float64 and sums.numel() >= 16
    vox = probs.numel() // (nsamples * n_labels)
    assert vox * nsamples * n_labels == probs.numel() == y_true.numel() and gsums.numel() >= 3 * nsamples * n_labels
    check(lib().fmri_weighted_dice_fwd(_p(probs), _p(y_true), _p(gsums), _p(sums), nsamples, vox, n_labels, float(smooth), _s()), "fmri_weighted_dice_fwd")


def weighted_dice_bwd(probs, y_true, gsums, sums, dlogits, nsamples, n_labels, smooth=WEIGHTED_DICE_SMOOTH, grad_scale=1.0):
    _need_cuda(probs, y_true, gsums, sums, dlogits)
    vox = probs.numel() // (nsamples * n_labels)
    check(lib().fmri_weighted_dice_bwd(_p(probs), _p(y_true), _p(gsums), _p(sums), _p(dlogits), nsamples, vox, n_labels, float(smooth),
                                       float(grad_scale), _s()), "fmri_weighted_dice_bwd")


def sigmoid_loss_bwd(probs, y_true, sums, dlogits, kind, param=1.0, smooth=1.0, grad_scale=1.0, weight=None):
    _need_cuda(probs, y_true, sums, dlogits, weight)
    if weight is not None:
        check(lib().fmri_sigmoid_loss_bwd_weighted(_p(probs), _p(y_true), _p(weight), _p(sums), _p(dlogits), probs.numel(), int(kind), float(param),
                                                   float(smooth), float(grad_scale), _s()), "fmri_sigmoid_loss_bwd_weighted")
        return
    check(lib().fmri_sigmoid_loss_bwd(_p(probs), _p(y_true), _p(sums), _p(dlogits), probs.numel(), int(kind), float(param), float(smooth),
                                      float(grad_scale), _s()), "fmri_sigmoid_loss_bwd")


def loss_value_from_sums(s, kind, param=1.0, smooth=1.0):
    """host-side value of the loss `kind` from the 16 metric sums (float64)"""
    I, Sy, Sp, n = float(s[0]), float(s[1]), float(s[2]), float(s[7])
    dice = (2 * I + smooth) / (Sy + Sp + smooth)
    if kind == 0:
        return -dice
    if kind == 1:
        return float(s[8]) / n
    if kind == 2:
        return -dice + param * float(s[8]) / n
    if kind == 3:
        return float(s[9])
    if kind == 4:
        return -(I + smooth) / (Sy + Sp - I + smooth)
    if kind == LOSS_WEIGHTED_DICE:
        return -float(s[10]) / float(s[11])          # mean over the (sample, label) groups of the (all-reduced) per-group Dice
    return -dice + param * (2 * (Sp - I) + smooth) / ((n - Sy) + Sp + smooth)


def clock_stamp(out16):
    """one {s_memtime, s_memrealtime} pair per XCD into out16 (int64[16], device) on the current stream (include/fmri_hip.h: fmri_clock_stamp)"""
    _need_cuda(out16)
    assert out16.dtype == torch.int64 and out16.numel() >= 16 and out16.is_contiguous()
    check(lib().fmri_clock_stamp(_p(out16), _s()), "fmri_clock_stamp")


def clock_ghz(stamp0, stamp1):
    """average shader clock between two stamps: median over the XCDs both stamps reached of d(memtime) / d(memrealtime) x 0.1 GHz;
    returns (GHz or None, per-XCD list)"""
    a, b = stamp0.cpu().numpy().astype("int64"), stamp1.cpu().numpy().astype("int64")
    per = []
    for x in range(8):
        dt, dr = int(b[2 * x] - a[2 * x]), int(b[2 * x + 1] - a[2 * x + 1])
        if a[2 * x + 1] != 0 and dr > 0 and dt > 0:
            per.append(dt / dr * 0.1)
    if not per:
        return None, per
    srt = sorted(per)
    return srt[len(srt) // 2], per


def set_deterministic(grad, shadow):
    """register (fp32 gradient buffer, zeroed int64 shadow of the same length) for bit-reproducible gradient accumulation, or (None, None)
    to switch it off (include/fmri_hip.h: fmri_set_deterministic); process-wide"""
    if grad is None:
        check(lib().fmri_set_deterministic(0, 0, 0), "fmri_set_deterministic")
        return
    _need_cuda(grad, shadow)
    assert grad.dtype == torch.float32 and shadow.dtype == torch.int64 and shadow.numel() == grad.numel() and grad.is_contiguous() and shadow.is_contiguous()
    check(lib().fmri_set_deterministic(_p(grad), _p(shadow), grad.numel()), "fmri_set_deterministic")


def deterministic_finish(grad, shadow):
    """grad += shadow * 2^-40, shadow = 0: once per backward pass, behind every gradient kernel"""
    _need_cuda(grad, shadow)
    check(lib().fmri_deterministic_finish(_p(grad), _p(shadow), grad.numel(), _s()), "fmri_deterministic_finish")


def conv3d_wgrad_workspace_bytes(C0, C1, Cout, N, D, H, W, dtype, planar=False):
    d = BF16 if dtype == torch.bfloat16 else F32
    return int(lib().fmri_conv3d_wgrad_workspace_bytes(C0, C1, Cout, N, D, H, W, d, int(planar)))


# ---------------------------------------------------------------------------------------------- patch sampler / intensity augmentation
U8 = 2


def _dt_any(t):
    return U8 if t.dtype == torch.uint8 else dt(t)


def affine_sample(vol, affine, start, size, out, order=1, cval=0.0, out_ld=None):
    """out[(i*ny + j)*out_ld + k] = vol sampled at affine . (start + (i,j,k), 1); `affine` is a host 3x4 / 4x4 array-like.
    `out` may be a view into a wider channels-last tensor (pass its row length as out_ld)."""
    import ctypes
    import numpy as np
    if not vol.is_cuda or not out.is_cuda or not vol.is_contiguous():
        raise RuntimeError("fmri_hip ops need device tensors (no CPU path)")
    X, Y, Z = vol.shape
    a = np.ascontiguousarray(np.asarray(affine, dtype=np.float64)[:3, :4]).reshape(12)
    arr = (ctypes.c_double * 12)(*a.tolist())
    nx, ny, nz = (int(v) for v in size)
    check(lib().fmri_affine_sample(_p(vol), _dt_any(vol), X, Y, Z, ctypes.cast(arr, ctypes.c_void_p), int(start[0]), int(start[1]), int(start[2]),
                                   nx, ny, nz, int(order), float(cval), _p(out), _dt_any(out), int(out_ld if out_ld is not None else nz), _s()),
          "fmri_affine_sample")
    return out


def minmax(x, out2):
    _need_cuda(x, out2)
    check(lib().fmri_minmax(_p(x), x.numel(), dt(x), _p(out2), _s()), "fmri_minmax")
    return out2


def rescale_intensity(x, stats, contrast, lo=0.0, hi=0.0, mult=1.0):
    _need_cuda(x, stats)
    check(lib().fmri_rescale_intensity(_p(x), x.numel(), dt(x), _p(stats), 1 if contrast else 0, float(lo), float(hi), float(mult), _s()),
          "fmri_rescale_intensity")
    return x


def noise_augment(x, stats, noise, kind, sigma):
    _need_cuda(x, stats, noise)
    assert noise.dtype == torch.float32 and noise.numel() == x.numel()
    check(lib().fmri_noise_augment(_p(x), x.numel(), dt(x), _p(stats), _p(noise), int(kind), float(sigma), _s()), "fmri_noise_augment")
    return x


# ---------------------------------------------------------------------------------------------- up-sample + concat + conv, parity form
def conv3d_upcat_ok(C0, C1, Cout, D, H, W, dtype, planar=False):
    """bit 0: forward / input gradients, bit 1: weight gradient.  planar: 2-D slices [1][S][H][W][C] (D = S is not up-sampled)"""
    f = lib().fmri_conv2d_upcat_ok if planar else lib().fmri_conv3d_upcat_ok
    return int(f(C0, C1, Cout, D, H, W, BF16 if dtype == torch.bfloat16 else F32))


def conv3d_pack_up_weights(w, C0, C1, up_f=None, up_d=None, sk_f=None, sk_d=None, planar=False):
    _need_cuda(w, up_f, up_d, sk_f, sk_d)
    Cout = w.shape[1]
    ref = next(t for t in (up_f, up_d, sk_f, sk_d) if t is not None)
    f = lib().fmri_conv2d_pack_up_weights if planar else lib().fmri_conv3d_pack_up_weights
    check(f(_p(w), C0, C1, Cout, _p(up_f), _p(up_d), _p(sk_f), _p(sk_d), dt(ref), _s()), "fmri_conv%dd_pack_up_weights" % (2 if planar else 3))


def conv3d_upcat_fwd(src0_low, src1, w_up_f, w_sk_f, bias, y, act=ACT_RELU, alpha=0.0, planar=False):
    _need_cuda(src0_low, src1, w_up_f, w_sk_f, bias, y)
    N, D, H, W, Cout = y.shape
    c0, c1 = src0_low.shape[-1], (0 if src1 is None else src1.shape[-1])
    if planar:
        assert N == 1
        check(lib().fmri_conv2d_upcat_fwd(_p(src0_low), c0, _p(src1), c1, _p(w_up_f), _p(w_sk_f), _p(bias), _p(y), D, H, W, Cout, act, float(alpha),
                                          dt(y), _s()), "fmri_conv2d_upcat_fwd")
        return y
    check(lib().fmri_conv3d_upcat_fwd(_p(src0_low), c0, _p(src1), c1, _p(w_up_f), _p(w_sk_f), _p(bias), _p(y), N, D, H, W,
                                      Cout, act, float(alpha), dt(y), _s()), "fmri_conv3d_upcat_fwd")
    return y


def conv3d_upcat_dgrad(dy, w_up_d, w_sk_d, mask_low, mask_skip, dx_low, dx_skip, planar=False):
    _need_cuda(dy, w_up_d, w_sk_d, mask_low, mask_skip, dx_low, dx_skip)
    N, D, H, W, Cout = dy.shape
    c0, c1 = dx_low.shape[-1], (0 if dx_skip is None else dx_skip.shape[-1])
    if planar:
        assert N == 1
        check(lib().fmri_conv2d_upcat_dgrad(_p(dy), Cout, _p(w_up_d), _p(w_sk_d), _p(mask_low), _p(mask_skip), _p(dx_low), _p(dx_skip), D, H, W,
                                            c0, c1, dt(dy), _s()), "fmri_conv2d_upcat_dgrad")
        return
    check(lib().fmri_conv3d_upcat_dgrad(_p(dy), Cout, _p(w_up_d), _p(w_sk_d), _p(mask_low), _p(mask_skip), _p(dx_low), _p(dx_skip), N, D, H, W,
                                        c0, c1, dt(dy), _s()), "fmri_conv3d_upcat_dgrad")


def conv3d_stride2_fwd(x, w_s2_fwd, bias, y):
    """Conv3D(3x3x3, strides 2, 'same') forward: x [N,D,H,W,Cin] (even dims), w_s2_fwd [8,8,Cout,Cin] (StridedParity.pack), bias fp32 [Cout] or
    None (added in the fp32 accumulators), y [N,D/2,H/2,W/2,Cout] (fmri_conv3d_stride2_fwd)"""
    _need_cuda(x, w_s2_fwd, bias, y)
    N, D, H, W, Cin = x.shape
    Cout = y.shape[-1]
    assert tuple(y.shape) == (N, D // 2, H // 2, W // 2, Cout) and tuple(w_s2_fwd.shape) == (8, 8, Cout, Cin)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() >= Cout)
    check(lib().fmri_conv3d_stride2_fwd(_p(x), Cin, _p(w_s2_fwd), _p(bias), _p(y), N, D, H, W, Cout, dt(x), _s()), "fmri_conv3d_stride2_fwd")
    return y


def conv3d_upcat_wgrad(src0_low, src1, dy, dw, db, dwc_scratch, workspace=None, planar=False):
    _need_cuda(src0_low, src1, dy, dw, db, dwc_scratch, workspace)
    N, D, H, W, Cout = dy.shape
    C0, C1 = src0_low.shape[-1], (0 if src1 is None else src1.shape[-1])
    assert dwc_scratch.dtype == torch.float32 and dwc_scratch.numel() >= (16 if planar else 64) * Cout * C0
    nws = 0 if workspace is None else workspace.numel() * workspace.element_size()
    if planar:
        assert N == 1
        check(lib().fmri_conv2d_upcat_wgrad(_p(src0_low), C0, _p(src1), C1, _p(dy), _p(dw), _p(db), _p(dwc_scratch), D, H, W, Cout, dt(dy),
                                            _p(workspace), nws, _s()), "fmri_conv2d_upcat_wgrad")
        return
    check(lib().fmri_conv3d_upcat_wgrad(_p(src0_low), C0, _p(src1), C1, _p(dy), _p(dw), _p(db), _p(dwc_scratch), N, D, H, W, Cout, dt(dy),
                                        _p(workspace), nws, _s()),
          "fmri_conv3d_upcat_wgrad")


# ---- Deconvolution3D -> concatenate -> Conv3D folded into one parity-form convolution (fmri_hip.deconv_fold holds the weight algebra)
def conv3d_upcat_fwd_bias27(src0_low, src1, w_up_f, w_sk_f, bias27, y, act=ACT_RELU, alpha=0.0):
    """as conv3d_upcat_fwd with the effective bias per border class of the output voxel: bias27 [27, Cout] fp32"""
    _need_cuda(src0_low, src1, w_up_f, w_sk_f, bias27, y)
    N, D, H, W, Cout = y.shape
    assert bias27.dtype == torch.float32 and tuple(bias27.shape) == (27, Cout) and bias27.is_contiguous()
    check(lib().fmri_conv3d_upcat_fwd_bias27(_p(src0_low), src0_low.shape[-1], _p(src1), src1.shape[-1], _p(w_up_f), _p(w_sk_f), _p(bias27), _p(y),
                                             N, D, H, W, Cout, act, float(alpha), dt(y), _s()), "fmri_conv3d_upcat_fwd_bias27")
    return y


def conv3d_upcat_wgrad_parts(src0_low, src1, dy, dw, db, dwc, workspace=None):
    """parity-filter gradients into dwc [8, 8, Cout, C0] fp32 (zeroed by the call), skip columns of dw [27, Cout, C0 + C1] and db accumulated"""
    _need_cuda(src0_low, src1, dy, dw, db, dwc, workspace)
    N, D, H, W, Cout = dy.shape
    C0, C1 = src0_low.shape[-1], src1.shape[-1]
    assert dwc.dtype == torch.float32 and dwc.numel() >= 64 * Cout * C0 and tuple(dw.shape) == (27, Cout, C0 + C1)
    nws = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(lib().fmri_conv3d_upcat_wgrad_parts(_p(src0_low), C0, _p(src1), C1, _p(dy), _p(dw), _p(db), _p(dwc), N, D, H, W, Cout, dt(dy), _p(workspace),
                                              nws, _s()), "fmri_conv3d_upcat_wgrad_parts")


def border_class_sums(dy, out27):
    """out27 [27, C] fp32 += per-border-class sums of dy [N, D, H, W, C]; the interior class (row 13) is left untouched"""
    _need_cuda(dy, out27)
    N, D, H, W, C = dy.shape
    assert out27.dtype == torch.float32 and tuple(out27.shape) == (27, C) and out27.is_contiguous()
    check(lib().fmri_border_class_sums(_p(dy), _p(out27), N, D, H, W, C, dt(dy), _s()), "fmri_border_class_sums")
    return out27


# ---------------------------------------------------------------------------------------------------- post-processing (fetal_net.postprocess)
def gaussian_filter_f64(vol, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(vol, sigma) (order 0, mode 'reflect') of a float64 device volume [X,Y,Z]: three separable passes with
    scipy's own weights and summation order - bit-identical to the host result"""
    import numpy as np
    _need_cuda(vol)
    assert vol.dtype == torch.float64 and vol.dim() == 3
    X, Y, Z = vol.shape
    sig = [float(sigma)] * 3 if np.isscalar(sigma) else [float(v) for v in sigma]
    a, b = vol, None
    for axis, sd in enumerate(sig):
        if sd <= 1e-15:                                       # scipy skips an axis whose sigma is ~0
            continue
        radius = int(truncate * sd + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sd * sd) * x ** 2)               # scipy.ndimage._filters._gaussian_kernel1d, order 0
        with torch.cuda.device(vol.device):                   # a volume on a non-current GPU: its device's stream, its device's weights
            w = torch.from_numpy(phi / phi.sum()).to(vol.device)
            b = torch.empty_like(vol)
            check(lib().fmri_correlate1d_f64(_p(a), _p(b), X, Y, Z, axis, _p(w), radius, _s()), "fmri_correlate1d_f64")
        a = b
    return a if a is not vol else vol.clone()


def threshold_f64(vol, thr):
    _need_cuda(vol)
    out = torch.empty(vol.shape, dtype=torch.uint8, device=vol.device)
    with torch.cuda.device(vol.device):
        check(lib().fmri_threshold_f64(_p(vol), _p(out), vol.numel(), float(thr), _s()), "fmri_threshold_f64")
    return out


def _until_stable(step, device, sweeps=16, limit=100000):
    """run `step(sweeps, changed)` until a batch of sweeps changes nothing (one 4-byte read-back per batch); the flag lives on the
    volume's device (the callers run under torch.cuda.device(volume.device), so _s() is that device's stream)"""
    changed = torch.zeros(1, dtype=torch.int32, device=device)
    done = 0
    while done < limit:
        changed.zero_()
        step(sweeps, changed)
        done += sweeps
        if int(changed.item()) == 0:
            return done
    raise RuntimeError("label propagation did not converge")


def binary_fill_holes_u8(mask):
    """scipy.ndimage.binary_fill_holes (default 6-connectivity) of a uint8 0/1 device volume"""
    _need_cuda(mask)
    assert mask.dtype == torch.uint8 and mask.dim() == 3
    X, Y, Z = mask.shape
    reached, out = torch.empty_like(mask), torch.empty_like(mask)
    L = lib()
    with torch.cuda.device(mask.device):
        check(L.fmri_fill_holes_step(_p(mask), _p(reached), 0, X, Y, Z, 0, 0, 0, _s()), "fmri_fill_holes_step")
        _until_stable(lambda n, ch: check(L.fmri_fill_holes_step(_p(mask), _p(reached), 0, X, Y, Z, 1, n, _p(ch), _s()), "fmri_fill_holes_step"),
                      mask.device)
        check(L.fmri_fill_holes_step(_p(mask), _p(reached), _p(out), X, Y, Z, 2, 0, 0, _s()), "fmri_fill_holes_step")
    return out


def largest_component_u8(mask):
    """mask of the largest 6-connected component (scipy.ndimage.label + argmax of the sizes; all zero when there is no foreground)"""
    _need_cuda(mask)
    assert mask.dtype == torch.uint8 and mask.dim() == 3
    X, Y, Z = mask.shape
    n = mask.numel()
    labels = torch.empty(n, dtype=torch.int32, device=mask.device)
    counts = torch.empty(n + 1, dtype=torch.int32, device=mask.device)
    best = torch.empty(1, dtype=torch.int64, device=mask.device)
    out = torch.empty_like(mask)
    L = lib()
    with torch.cuda.device(mask.device):
        check(L.fmri_largest_component_step(_p(mask), _p(labels), 0, 0, 0, X, Y, Z, 0, 0, 0, _s()), "fmri_largest_component_step")
        _until_stable(lambda k, ch: check(L.fmri_largest_component_step(0, _p(labels), 0, 0, 0, X, Y, Z, 1, k, _p(ch), _s()),
                                          "fmri_largest_component_step"), mask.device, sweeps=8)
        check(L.fmri_largest_component_step(0, _p(labels), _p(counts), _p(best), _p(out), X, Y, Z, 2, 0, 0, _s()), "fmri_largest_component_step")
    return out


# ---------------------------------------------------------------------------------------------------- discriminator head (discriminator.hip)
def avgpool_fwd(x, y, planar=False):
    """x [N,D,H,W,C] -> y [N,D//2,H//2,W//2,C] (planar: [N,D,H//2,W//2,C]); AveragePooling3D() / AveragePooling2D()"""
    _need_cuda(x, y)
    N, D, H, W, Cc = x.shape
    assert tuple(y.shape) == (N, D if planar else D // 2, H // 2, W // 2, Cc), (x.shape, y.shape)
    check(lib().fmri_avgpool3d_2x_fwd(_p(x), _p(y), N, D, H, W, Cc, dt(x), int(planar), _s()), "fmri_avgpool3d_2x_fwd")
    return y


def avgpool_bwd(dy, dx, planar=False):
    _need_cuda(dy, dx)
    N, D, H, W, Cc = dx.shape
    assert tuple(dy.shape) == (N, D if planar else D // 2, H // 2, W // 2, Cc), (dx.shape, dy.shape)
    check(lib().fmri_avgpool3d_2x_bwd(_p(dy), _p(dx), N, D, H, W, Cc, dt(dx), int(planar), _s()), "fmri_avgpool3d_2x_bwd")
    return dx


def global_avgpool_fwd(x, y):
    """x [N, ..., C] -> y [N, C] fp32"""
    _need_cuda(x, y)
    N, Cc = x.shape[0], x.shape[-1]
    assert y.dtype == torch.float32 and tuple(y.shape) == (N, Cc)
    check(lib().fmri_global_avgpool_fwd(_p(x), _p(y), N, x.numel() // (N * Cc), Cc, dt(x), _s()), "fmri_global_avgpool_fwd")
    return y


def global_avgpool_bwd(dy, dx):
    _need_cuda(dy, dx)
    N, Cc = dx.shape[0], dx.shape[-1]
    assert dy.dtype == torch.float32 and tuple(dy.shape) == (N, Cc)
    check(lib().fmri_global_avgpool_bwd(_p(dy), _p(dx), N, dx.numel() // (N * Cc), Cc, dt(dx), _s()), "fmri_global_avgpool_bwd")
    return dx


def dense_fwd(x, w, b, y, act=ACT_NONE, alpha=0.0):
    """x [N,K] fp32, w [K,M] (Keras kernel), b [M] -> y [N,M] fp32"""
    _need_cuda(x, w, b, y)
    N, K = x.shape
    M = w.shape[1]
    assert w.shape[0] == K and tuple(y.shape) == (N, M) and all(t.dtype == torch.float32 for t in (x, w, y))
    check(lib().fmri_dense_fwd(_p(x), _p(w), _p(b), _p(y), N, K, M, act, float(alpha), _s()), "fmri_dense_fwd")
    return y


def dense_bwd(x, w, y, dy, dx, dw, db, act=ACT_NONE, alpha=0.0):
    """dw / db accumulate, dx is written; any of the three may be None"""
    _need_cuda(x, w, y, dy, dx, dw, db)
    N, K = x.shape
    M = w.shape[1]
    check(lib().fmri_dense_bwd(_p(x), _p(w), _p(y), _p(dy), _p(dx), _p(dw), _p(db), N, K, M, act, float(alpha), _s()), "fmri_dense_bwd")


def sigmoid_bce_fwd(logits, target, probs, sums):
    """sums (fp64, >= 3) += [sum of Keras binary_crossentropy terms, sum |p - t|, n]"""
    _need_cuda(logits, target, probs, sums)
    assert target.dtype == torch.float32 and logits.dtype == torch.float32 and sums.dtype == torch.float64
    check(lib().fmri_sigmoid_bce_fwd(_p(logits), _p(target), _p(probs), _p(sums), logits.numel(), _s()), "fmri_sigmoid_bce_fwd")
    return probs


def sigmoid_bce_bwd(probs, target, dlogits, scale):
    _need_cuda(probs, target, dlogits)
    check(lib().fmri_sigmoid_bce_bwd(_p(probs), _p(target), _p(dlogits), probs.numel(), float(scale), _s()), "fmri_sigmoid_bce_bwd")
    return dlogits


def sigmoid_chain(probs, dprobs, dlogits, scale=1.0, accumulate=False):
    """dlogits [nvox, L] (+)= scale * dprobs[..., :L] * p * (1 - p); dprobs [..., ld] is the discriminator's input gradient"""
    _need_cuda(probs, dprobs, dlogits)
    nvox, L = probs.shape
    ld = dprobs.shape[-1]
    assert dprobs.numel() == nvox * ld and dlogits.shape == probs.shape
    check(lib().fmri_sigmoid_chain(_p(probs), _p(dprobs), ld, L, _p(dlogits), nvox, float(scale), int(accumulate), dt(dprobs), _s()),
          "fmri_sigmoid_chain")
    return dlogits


def discriminator_input(probs, x, out, merge=False):
    """out [..., ld] = [probs, x, 0...] or, merge, the mul-merge maps [x * probs, x * (1 - probs), 0...] (train_adv.py:92-95)"""
    _need_cuda(probs, x, out)
    nvox, L = probs.shape
    Cc = x.shape[-1]
    assert x.numel() == nvox * Cc and out.numel() == nvox * out.shape[-1]
    check(lib().fmri_discriminator_input(_p(probs), L, _p(x), Cc, dt(x), _p(out), out.shape[-1], dt(out), nvox, int(merge), _s()),
          "fmri_discriminator_input")
    return out


# ---------------------------------------------------------------------------------------------------- augmenters of the skimage family
def gaussian_filter_f32(patch, sigma, truncate=4.0, mode="nearest"):
    """skimage.filters.gaussian(patch, sigma) of an fp32 device patch [X,Y,Z] (reference augment.py:113-114): scipy's weights, mode
    'nearest'; a patch with 3 planes along the last axis is not smoothed along it (skimage takes it for an RGB image).  Returns a new
    tensor (or `patch` itself when every sigma is ~0)."""
    import numpy as np
    _need_cuda(patch)
    assert patch.dtype == torch.float32 and patch.dim() == 3
    X, Y, Z = patch.shape
    if np.isscalar(sigma):
        sig = [float(sigma)] * 3
        if Z == 3:
            sig[2] = 0.0
    else:                                                 # one sigma per axis (0: the axis is left alone)
        sig = [float(v) for v in sigma]
        assert len(sig) == 3
    a = patch
    for axis, sd in enumerate(sig):
        if sd <= 1e-15:
            continue
        radius = int(truncate * sd + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sd * sd) * x ** 2)
        w = torch.from_numpy(phi / phi.sum()).to(patch.device)
        b = torch.empty_like(patch)
        check(lib().fmri_correlate1d_f32(_p(a), _p(b), X, Y, Z, axis, _p(w), radius, 1 if mode == "nearest" else 0, _s()), "fmri_correlate1d_f32")
        a = b
    return a


def elastic_ksize(sigma):
    """imgaug 0.4.0 blur.py `_compute_gaussian_blur_ksize` + the odd-size rule of `blur_gaussian_`: width of the truncated Gaussian kernel"""
    k = 3.3 * sigma if sigma < 3.0 else (2.9 * sigma if sigma < 5.0 else 2.6 * sigma)
    k = int(max(k, 5))
    return k + 1 if k % 2 == 0 else k


_ELASTIC_KERNELS = {}
ELASTIC_RNG_KMAX = 31


def _elastic_kernel(k, sigma, device):
    """the truncated, normalised Gaussian of cv2.GaussianBlur(ksize=k, sigma) as a device fp64 vector; one pageable upload (a host sync) per
    (k, sigma, device), not per patch"""
    import numpy as np
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (k, float(sigma), device)
    wd = _ELASTIC_KERNELS.get(key)
    if wd is None:
        xs = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
        w = np.exp(-(xs * xs) / (2.0 * float(sigma) ** 2))
        wd = _ELASTIC_KERNELS[key] = torch.from_numpy(w / w.sum()).to(device)
    return wd


def elastic_fields_rng(shape2d, alpha, sigma, seed, seq, device="cuda"):
    """elastic_fields with the noise drawn in the kernel, one patch (see elastic_fields_rng_batch) -> (d0, d1)"""
    d = elastic_fields_rng_batch(shape2d, [alpha], sigma, seed, [seq], device=device)
    return d[0, 0], d[0, 1]


def elastic_fields(shape2d, alpha, sigma, generator=None, noise=None):
    """-> (d0, d1): the displacement along axis 0 (imgaug's dy) and axis 1 (dx) of an X x Y image, fp32 device tensors.
    imgaug 0.4.0 ElasticTransformation._generate_shift_maps: uniform(-1, 1) noise on the image padded by the kernel width on every side
    (one draw of (2 * h_pad, w_pad): dx from the first half, dy from the second), blurred with the truncated normalised Gaussian kernel
    (cv2.GaussianBlur(ksize, sigma)), times alpha, padding cropped - the border mode never reaches the kept part.  `noise`: the draw itself
    (fp32 device tensor (2, h_pad, w_pad), block 0 = dx) instead of the device generator's (tests feed the oracle's)."""
    import numpy as np
    X, Y = int(shape2d[0]), int(shape2d[1])
    k = elastic_ksize(float(sigma))
    hp, wp = X + 2 * k, Y + 2 * k
    if noise is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if generator is None else generator.device
        noise = torch.rand((2, hp, wp), device=dev, dtype=torch.float32, generator=generator) * 2 - 1
    _need_cuda(noise)
    assert tuple(noise.shape) == (2, hp, wp) and noise.dtype == torch.float32 and noise.is_contiguous()
    wd = _elastic_kernel(k, sigma, noise.device)
    a = noise
    for axis in (1, 2):
        b = torch.empty_like(a)
        check(lib().fmri_correlate1d_f32(_p(a), _p(b), 2, hp, wp, axis, _p(wd), k // 2, 0, _s()), "fmri_correlate1d_f32")
        a = b
    f = a[:, k:k + X, k:k + Y] * float(alpha)
    return f[1].contiguous(), f[0].contiguous()


def elastic_warp(src, d0, d1, order, out):
    """out[i, j, c] = src[:, :, c] at (i - d0[i, j], j - d1[i, j]), order 0 / 1, mode 'nearest' (fmri_elastic_warp); src, out: (X, Y, C) float32 or
    uint8 device tensors whose last axis may be a view into a wider row (stride(1) = row length)."""
    _need_cuda(d0, d1)
    if not src.is_cuda or not out.is_cuda:
        raise RuntimeError("fmri_hip ops need device tensors (no CPU path)")
    X, Y, C = src.shape
    assert tuple(out.shape) == (X, Y, C) and out.dtype == src.dtype and tuple(d0.shape) == (X, Y) == tuple(d1.shape)
    assert d0.dtype == torch.float32 and d1.dtype == torch.float32
    for t in (src, out):
        assert t.stride(2) == 1 and t.stride(0) == Y * t.stride(1), "rows of equal length, channels contiguous"
    check(lib().fmri_elastic_warp(_p(src), _dt_any(src), X, Y, C, int(src.stride(1)), _p(d0), _p(d1), int(order), _p(out), int(out.stride(1)), _s()),
          "fmri_elastic_warp")
    return out


def piecewise_affine_matrices(shape2d, dest_yx):
    """the two triangle maps of imgaug PiecewiseAffine(nb_rows=2, nb_cols=2): corners (0,0), (0,w), (h,0), (h,w) (y, x) moved to dest_yx (4, 2);
    triangle 0 = {p0, p2, p3}: in = d0 + (y/h)(d2 - d0) + (x/w)(d3 - d2); triangle 1 = {p0, p1, p3}: in = d0 + (x/w)(d1 - d0) + (y/h)(d3 - d1).
    -> 12 floats: per triangle (row_in = a i + b j + c), (col_in = a i + b j + c)"""
    import numpy as np
    h, w = float(shape2d[0]), float(shape2d[1])
    d = np.asarray(dest_yx, dtype=np.float64)
    out = []
    for (dy, dx) in (((d[2] - d[0]) / h, (d[3] - d[2]) / w), ((d[3] - d[1]) / h, (d[1] - d[0]) / w)):
        out += [dy[0], dx[0], d[0][0], dy[1], dx[1], d[0][1]]
    return np.asarray(out, dtype=np.float64)


def piecewise_affine(src, dest_yx, order, out):
    """out[i, j, c] = src[:, :, c] at the piecewise-affine image of (i, j) (fmri_piecewise_affine2); dest_yx: the moved corners, host (4, 2)"""
    import ctypes
    if not src.is_cuda or not out.is_cuda:
        raise RuntimeError("fmri_hip ops need device tensors (no CPU path)")
    X, Y, C = src.shape
    assert tuple(out.shape) == (X, Y, C) and out.dtype == src.dtype
    for t in (src, out):
        assert t.stride(2) == 1 and t.stride(0) == Y * t.stride(1), "rows of equal length, channels contiguous"
    m = piecewise_affine_matrices((X, Y), dest_yx)
    arr = (ctypes.c_double * 12)(*m.tolist())
    check(lib().fmri_piecewise_affine2(_p(src), _dt_any(src), X, Y, C, int(src.stride(1)), ctypes.cast(arr, ctypes.c_void_p), int(order), _p(out),
                                       int(out.stride(1)), _s()), "fmri_piecewise_affine2")
    return out


def coarse_dropout(x, keep, stats, per_channel=True):
    """in place on x (X, Y, C) fp32 / bf16 (last axis may be a view into a wider row): voxels whose cell of `keep` (uint8 (hs, ws, C) or (hs, ws, 1))
    is 0 take stats[0] (= minmax(x) before the call)"""
    _need_cuda(keep, stats)
    X, Y, C = x.shape
    hs, ws, kc = keep.shape
    assert keep.dtype == torch.uint8 and keep.is_contiguous() and kc == (C if per_channel else 1)
    assert x.stride(2) == 1 and x.stride(0) == Y * x.stride(1)
    check(lib().fmri_coarse_dropout(_p(x), dt(x), X, Y, C, int(x.stride(1)), _p(keep), hs, ws, kc, _p(stats), _s()), "fmri_coarse_dropout")
    return x


AUG_WS_INTS = 1568        # FMRI_AUG_WS_INTS
_U64 = 2 ** 64 - 1


def aug_workspace(device, batch=1):
    """(stats, ws) for the *_rng intensity steps: [batch][2] floats and the zeroed int32 workspaces they keep re-armed; one pair per stream of
    calls.  The single-patch wrappers below take row 0 of a batch-1 workspace as (stats, ws)."""
    stats = torch.zeros((batch, 2), device=device, dtype=torch.float32)
    ws = torch.zeros((batch, AUG_WS_INTS), device=device, dtype=torch.int32)
    return (stats[0], ws[0]) if batch == 1 else (stats, ws)


def _batch_geometry(xb):
    """xb: (B, ...) device tensor whose patches are dense and equally spaced -> (B, elements per patch, stride in elements)"""
    B = xb.shape[0]
    n = xb[0].numel()
    assert xb.is_cuda and xb[0].is_contiguous() and (B == 1 or xb.stride(0) >= n)
    return B, n, (xb.stride(0) if B > 1 else n)


def _seqs(seqs):
    import numpy as np
    a = np.ascontiguousarray(seqs, dtype=np.uint32)
    return a, a.ctypes.data


def minmax_ws_batch(xb, stats, ws):
    """stats[b] = {min, max} of xb[b], all patches in one launch (fmri_minmax_ws_batch)"""
    _need_cuda(stats, ws)
    B, n, stride = _batch_geometry(xb)
    check(lib().fmri_minmax_ws_batch(_p(xb), n, stride, B, dt(xb), _p(stats), _p(ws), _s()), "fmri_minmax_ws_batch")
    return stats


def rescale_intensity_ws_batch(xb, stats, ws, params):
    """params: (B, 4) rows {mode, lo, hi, mult}, mode 0 skip / 1 multiply / 2 contrast + multiply; stats[b] updated"""
    import numpy as np
    _need_cuda(stats, ws)
    B, n, stride = _batch_geometry(xb)
    a = np.ascontiguousarray(params, dtype=np.float32)
    assert a.shape == (B, 4)
    check(lib().fmri_rescale_intensity_ws_batch(_p(xb), n, stride, B, dt(xb), _p(stats), _p(ws), a.ctypes.data, _s()), "fmri_rescale_intensity_ws_batch")
    return xb


def noise_rng_batch(xb, stats, ws, kind, sigma, seed, seqs):
    """gaussian (kind 0) / speckle (1) noise in place on every patch with seqs[b] != 0, the normal draws in the kernel; stats[b] updated"""
    _need_cuda(stats, ws)
    B, n, stride = _batch_geometry(xb)
    a, ap = _seqs(seqs)
    assert a.shape == (B,)
    check(lib().fmri_noise_rng_batch(_p(xb), n, stride, B, dt(xb), _p(stats), _p(ws), int(kind), float(sigma), int(seed) & _U64, ap, _s()),
          "fmri_noise_rng_batch")
    return xb


def shot_noise_rng_batch(xb, stats, ws, seed, seqs):
    """reference augment.py:87-94 in place on every patch with seqs[b] != 0, the Poisson draws in the kernel; stats[b] updated"""
    _need_cuda(stats, ws)
    B, n, stride = _batch_geometry(xb)
    a, ap = _seqs(seqs)
    assert a.shape == (B,)
    check(lib().fmri_shot_noise_rng_batch(_p(xb), n, stride, B, dt(xb), _p(stats), _p(ws), int(seed) & _U64, ap, _s()), "fmri_shot_noise_rng_batch")
    return xb


def coarse_dropout_rng_batch(xb, grids, rate, stats, per_channel, seed, seqs):
    """coarse dropout in place on xb (B, X, Y, C); grids: (B, 2) keep-grid sizes (per slice when per_channel), drawn in the kernel"""
    import numpy as np
    _need_cuda(stats)
    B, X, Y, C = xb.shape
    assert xb.is_cuda and xb.stride(3) == 1 and xb.stride(1) == Y * xb.stride(2)
    g = np.ascontiguousarray(grids, dtype=np.int32)
    a, ap = _seqs(seqs)
    assert g.shape == (B, 2) and a.shape == (B,)
    check(lib().fmri_coarse_dropout_rng_batch(_p(xb), dt(xb), X, Y, C, int(xb.stride(2)), int(xb.stride(0)), B, g.ctypes.data, C if per_channel else 1,
                                              float(rate), _p(stats), int(seed) & _U64, ap, _s()), "fmri_coarse_dropout_rng_batch")
    return xb


def affine_sample_batch(vols, affines, corners, size, out, order, cvals):
    """out[b] (dense (nx, ny, nz) patches, equally spaced) = vols[b] sampled at affines[b] . (corners[b] + (i, j, k), 1): fmri_affine_sample for the
    patches of a batch in one launch.  vols: device tensors of one dtype (float32 or uint8), each (X, Y, Z) contiguous; affines (B, 4, 4) or (B, 3, 4)"""
    import ctypes
    import numpy as np
    B = len(vols)
    nx, ny, nz = (int(v) for v in size)
    assert out.is_cuda and tuple(out.shape) == (B, nx, ny, nz) and out[0].is_contiguous()
    for v in vols:
        if not v.is_cuda or not v.is_contiguous() or v.dtype != vols[0].dtype or v.dim() != 3:
            raise RuntimeError("fmri_hip ops need contiguous device volumes of one dtype")
    ptrs = (ctypes.c_void_p * B)(*[v.data_ptr() for v in vols])
    dims = np.ascontiguousarray([v.shape for v in vols], dtype=np.int32)
    A = np.ascontiguousarray(np.asarray(affines, dtype=np.float64)[:, :3, :4])
    cr = np.ascontiguousarray(corners, dtype=np.int32)
    cv = np.ascontiguousarray(cvals, dtype=np.float32)
    assert A.shape == (B, 3, 4) and cr.shape == (B, 3) and cv.shape == (B,)
    check(lib().fmri_affine_sample_batch(B, ctypes.cast(ptrs, ctypes.c_void_p), dims.ctypes.data, A.ctypes.data, cr.ctypes.data, cv.ctypes.data,
                                         _dt_any(vols[0]), nx, ny, nz, int(order), _p(out), _dt_any(out), nz, int(out.stride(0)), _s()),
          "fmri_affine_sample_batch")
    return out


def elastic_fields_rng_batch(shape2d, alphas, sigma, seed, seqs, device="cuda"):
    """-> d (B, 2, X, Y) fp32: d[b, 0] = shift along axis 0, d[b, 1] along axis 1 (ops.elastic_fields' (d0, d1)), the noise drawn in the kernel
    (fmri_elastic_fields_rng_batch: one launch); seqs[b] == 0 -> zero fields.  Kernel widths above ELASTIC_RNG_KMAX (sigma >= 12) are not
    covered - callers use elastic_fields there"""
    import numpy as np
    X, Y = int(shape2d[0]), int(shape2d[1])
    k = elastic_ksize(float(sigma))
    assert k <= ELASTIC_RNG_KMAX
    wd = _elastic_kernel(k, sigma, device)
    al = np.ascontiguousarray(alphas, dtype=np.float32)
    a, ap = _seqs(seqs)
    B = al.shape[0]
    assert a.shape == (B,)
    d = torch.empty((B, 2, X, Y), device=wd.device, dtype=torch.float32)
    check(lib().fmri_elastic_fields_rng_batch(_p(d), X, Y, k, _p(wd), al.ctypes.data, int(seed) & _U64, ap, B, _s()), "fmri_elastic_fields_rng_batch")
    return d


def elastic_warp_batch(src, d, order, out):
    """out[b] = src[b] warped by the fields d[b] (elastic_fields_rng_batch); src, out: (B, X, Y, C) float32 or uint8, patches dense / equally spaced"""
    _need_cuda(d)
    B, X, Y, C = src.shape
    assert src.is_cuda and out.is_cuda and tuple(out.shape) == (B, X, Y, C) and out.dtype == src.dtype and tuple(d.shape) == (B, 2, X, Y)
    assert d.is_contiguous() and d.dtype == torch.float32
    for t in (src, out):
        assert t.stride(3) == 1 and t.stride(1) == Y * t.stride(2), "rows of equal length, channels contiguous"
    check(lib().fmri_elastic_warp_batch(_p(src), _dt_any(src), X, Y, C, int(src.stride(2)), int(src.stride(0)), _p(d), int(order), _p(out),
                                        int(out.stride(2)), int(out.stride(0)), B, _s()), "fmri_elastic_warp_batch")
    return out


# one patch = a batch of one (tests; the generator's per-patch path)
def minmax_ws(x, stats, ws):
    minmax_ws_batch(x.unsqueeze(0), stats, ws)
    return stats


def rescale_intensity_ws(x, stats, ws, contrast, lo=0.0, hi=0.0, mult=1.0):
    rescale_intensity_ws_batch(x.unsqueeze(0), stats, ws, [[2 if contrast else 1, lo, hi, mult]])
    return x


def noise_rng(x, stats, ws, kind, sigma, seed, seq):
    noise_rng_batch(x.unsqueeze(0), stats, ws, kind, sigma, seed, [seq])
    return x


def shot_noise_rng(x, stats, ws, seed, seq):
    shot_noise_rng_batch(x.unsqueeze(0), stats, ws, seed, [seq])
    return x


def coarse_dropout_rng(x, grid, rate, stats, per_channel, seed, seq):
    coarse_dropout_rng_batch(x.unsqueeze(0), [[int(grid[0]), int(grid[1])]], rate, stats, per_channel, seed, [seq])
    return x


def shot_noise(x, stats, generator=None, draws_fn=None):
    """reference augment.py:87-94 in place on `x` (fp32 / bf16 device tensor); `stats` = minmax(x) taken before the call.  The Poisson
    draws come from torch's device generator (`draws_fn(rates)` overrides them: tests feed the oracle's own draws)."""
    _need_cuda(x, stats)
    n = x.numel()
    present = torch.zeros(1024, dtype=torch.int32, device=x.device)
    rates = torch.empty(n, dtype=torch.float32, device=x.device)
    L = lib()
    check(L.fmri_shot_noise_step(_p(x), n, _dt_any(x), _p(stats), _p(present), 0, 0, 0, _s()), "fmri_shot_noise_step(0)")
    check(L.fmri_shot_noise_step(_p(x), n, _dt_any(x), _p(stats), _p(present), _p(rates), 0, 1, _s()), "fmri_shot_noise_step(1)")
    draws = draws_fn(rates) if draws_fn is not None else torch.poisson(rates, generator=generator)
    draws = draws.to(torch.float32).contiguous()
    check(L.fmri_shot_noise_step(_p(x), n, _dt_any(x), _p(stats), _p(present), 0, _p(draws), 2, _s()), "fmri_shot_noise_step(2)")
    return rates
