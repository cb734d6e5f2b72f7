"""Weight algebra of `Deconvolution3D(k 2, s 2) -> concatenate([up, skip]) -> Conv3D(3x3x3)` folded into one parity-form convolution of
the low-res tensor (reference fetal_net/model/unet3d/unet.py:132-138, :61, :102 with deconvolution=True; kernels: fmri_conv3d_upcat_fwd_bias27,
fmri_conv3d_upcat_dgrad, fmri_conv3d_upcat_wgrad_parts, fmri_border_class_sums).

Per axis, output voxel 2g+p of the transposed conv is Wt[p] x[g] + bt.  Tap k (0, 1, 2) of the following conv at output voxel v = 2g+p reads
voxel v + k - 1 of the transposed conv's output = low-res voxel g + floor((p+k-1)/2) through Wt[a], a = (p+k-1) mod 2:

    p = 0:  k = 0 -> (g-1, a=1)   k = 1 -> (g, a=0)   k = 2 -> (g,   a=1)        combined tap t' = 0 for g-1, 1 for g
    p = 1:  k = 0 -> (g,   a=0)   k = 1 -> (g, a=1)   k = 2 -> (g+1, a=0)        combined tap t' = 0 for g,   1 for g+1

which is the neighbourhood of the nearest-upsample parity form (csrc/conv3d_api.hip, tap_class) with pre-MULTIPLIED filters

    Weff[p][t'] = sum over the taps k of class t' under p of  W3[k][:, :Cmid] @ Wt[a(p,k)]          [Cout, Cin]

The transposed conv's bias bt reaches an output voxel through the IN-VOLUME taps only (the conv zero-pads the transposed conv's output), so
the effective bias depends on the border class c = 0 (first voxel of the axis: tap 0 missing) / 1 (interior) / 2 (last: tap 2 missing) of
the output voxel per axis:  bias27[c] = b3 + sum_{k in K(c)} W3[k][:, :Cmid] @ bt.

Everything here is small dense algebra on the layer's parameters (216 GEMMs of Cout x Cmid x Cin per refresh) - torch matmuls on the device,
fp32; the convolutions themselves run on the MFMA kernels."""
import numpy as np
import torch


def _tables():
    tq = np.zeros((8, 27), np.int64)          # (parity, tap) -> combined tap index t' = (td*2 + th)*2 + tw
    aq = np.zeros((8, 27), np.int64)          # (parity, tap) -> transposed-conv tap a = (ad*2 + ah)*2 + aw
    cls = lambda p, k: (1 if k >= 1 else 0) if p == 0 else (1 if k >= 2 else 0)
    for p in range(8):
        pp = (p >> 2, (p >> 1) & 1, p & 1)
        for k in range(27):
            kk = (k // 9, (k // 3) % 3, k % 3)
            t = [cls(pp[i], kk[i]) for i in range(3)]
            a = [(pp[i] + kk[i] + 1) & 1 for i in range(3)]
            tq[p, k] = (t[0] * 2 + t[1]) * 2 + t[2]
            aq[p, k] = (a[0] * 2 + a[1]) * 2 + a[2]
    # border class (cd, ch, cw) x tap (kd, kh, kw): tap present?
    present = np.zeros((27, 27), np.float32)
    ok = lambda c, k: not ((c == 0 and k == 0) or (c == 2 and k == 2))
    for c in range(27):
        cc = (c // 9, (c // 3) % 3, c % 3)
        for k in range(27):
            kk = (k // 9, (k // 3) % 3, k % 3)
            present[c, k] = float(all(ok(cc[i], kk[i]) for i in range(3)))
    return tq, aq, present


class DeconvFold(object):
    """index tables on the device + the three pieces of algebra: effective filters, their gradients chained back, bias classes"""

    def __init__(self, device, gemm_dtype=None):
        """gemm_dtype: operand type of the weight GEMMs (None: the parameters' own, fp32; torch.bfloat16: operands rounded to bf16, products
        accumulated in fp32 by the GEMM library - the effective filters are rounded to bf16 for the MFMA kernels anyway)"""
        self.gemm_dtype = gemm_dtype
        tq, aq, present = _tables()
        self.tq = torch.from_numpy(tq).to(device)                      # [8, 27]
        self.aq = torch.from_numpy(aq).to(device)
        self.flat_t = (torch.arange(8, device=device)[:, None] * 8 + self.tq).reshape(-1)      # (p, k) -> p * 8 + t'
        self.present = torch.from_numpy(present).to(device)            # [27 classes, 27 taps]
        # every (tap k, transposed-conv tap a) pair occurs under exactly ONE parity p = (a + k + 1) mod 2 per axis: (k, a) -> p * 8 + t'(p, k)
        ka = np.zeros((27, 8), np.int64)
        for p_ in range(8):
            for k_ in range(27):
                ka[k_, aq[p_, k_]] = p_ * 8 + tq[p_, k_]
        self.flat_ka = torch.from_numpy(ka.reshape(-1)).to(device)      # [27 * 8]
        # w_up_dgrad image = Wc[p][1 - t']^T: flip of the three combined-tap axes
        self.mirror = torch.tensor([((t >> 2) ^ 1) * 4 + (((t >> 1) & 1) ^ 1) * 2 + ((t & 1) ^ 1) for t in range(8)], device=device)

    def effective(self, w3, wt, b3, bt, cmid, gemm_dtype=None):
        """w3 [27, Cout, Cmid + Cskip], wt [8, Cmid, Cin], b3 [Cout], bt [Cmid] (fp32) ->
        weff [8, 8, Cout, Cin], bias27 [27, Cout]  (fp32)"""
        w3u = w3[:, :, :cmid]                                           # [27, Cout, Cmid]
        gd = gemm_dtype or self.gemm_dtype or w3.dtype
        # all 27 x 8 products W3u[k] @ Wt[a] in one batched GEMM (no operand is gathered), scattered onto their (parity, combined tap)
        prod = torch.matmul(w3u.to(gd)[:, None], wt.to(gd)[None])       # [27, 8, Cout, Cin]
        Cout, Cin = prod.shape[-2:]
        weff = torch.zeros((64, Cout, Cin), dtype=w3.dtype, device=w3.device)
        weff.index_add_(0, self.flat_ka, prod.reshape(27 * 8, Cout, Cin).to(w3.dtype))
        wb = torch.matmul(w3u, bt)                                      # [27, Cout]: tap k's share of the transposed conv's bias
        bias27 = b3[None] + self.present.to(wb.dtype) @ wb
        return weff.view(8, 8, Cout, Cin), bias27.contiguous()

    def chain(self, dweff, w3, wt, bt, cmid, s27, gemm_dtype=None):
        """dweff [8, 8, Cout, Cin]: gradient w.r.t. the effective filters; s27 [27, Cout]: per-border-class sums of dy (all 27 rows) ->
        (dw3u [27, Cout, Cmid], dwt [8, Cmid, Cin], dbt [Cmid]); db3 = s27.sum(0) is the caller's (the kernels accumulate it themselves)"""
        w3u = w3[:, :, :cmid]
        gd = gemm_dtype or self.gemm_dtype or w3.dtype
        Cout, Cin = dweff.shape[-2:]
        # G[k, a] = gradient of the (parity, combined tap) block the pair (k, a) contributes to              [27, 8, Cout, Cin]
        g = dweff.reshape(64, Cout, Cin).to(gd)[self.flat_ka].view(27, 8, Cout, Cin)
        # dW3u[k] = sum_a G[k, a] @ Wt[a]^T: one GEMM per tap with the (a, ci) pairs as its contraction axis
        dw3u = torch.matmul(g.permute(0, 2, 1, 3).reshape(27, Cout, 8 * Cin), wt.to(gd).permute(0, 2, 1).reshape(8 * Cin, cmid)).to(w3.dtype)
        # dWt[a] = sum_k W3u[k]^T @ G[k, a]: one GEMM per transposed-conv tap with the (k, co) pairs as its contraction axis
        dwt = torch.matmul(w3u.to(gd).permute(2, 0, 1).reshape(cmid, 27 * Cout)[None],
                           g.permute(1, 0, 2, 3).reshape(8, 27 * Cout, Cin)).to(wt.dtype)
        # bias: L depends on bt through bias27[c] = ... + sum_{k in K(c)} W3u[k] bt  ->  dbt = sum_c sum_{k in K(c)} W3u[k]^T s27[c]
        sk = self.present.t().to(s27.dtype) @ s27                                                                 # [27 taps, Cout]
        dbt = torch.einsum("koc,ko->c", w3u, sk)
        # the part of dw3u that comes through the bias term: d/dW3u[k] of sum_c s27[c] . (W3u[k] bt) 1[k in K(c)] = sk[k] (outer) bt
        dw3u = dw3u + sk[:, :, None] * bt[None, None, :]
        return dw3u, dwt, dbt
