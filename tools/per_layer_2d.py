#!/usr/bin/env python3
"""Exclusive (one-stream) time of every op of the configs[3] 2-D training step (64 x 256x256x5, depth 4 / 32 filters, bf16), with the
algorithmic MFMA fraction of the 3x3 convs (2*9*Cin*Cout*pixels / time / 2.5 PF).  Tuning aid: python tools/per_layer_2d.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
import torch
from fmri_hip import ops
from fmri_hip.engine import UNetEngine, UNetPlan

rec = {}
on = [False]


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        if not on[0]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        shp = tuple(tuple(t.shape) if torch.is_tensor(t) else None for t in a[:7])
        rec.setdefault((name, shp), []).append((e0, e1))
        return r
    setattr(ops, name, f)


for n in ("conv3d_fwd", "conv3d_fwd_tail", "conv3d_dgrad", "conv3d_wgrad", "conv3d_upcat_fwd", "conv3d_upcat_dgrad", "conv3d_upcat_wgrad", "maxpool_fwd",
          "maxpool_bwd", "upsample_bwd", "conv1x1_fwd", "conv1x1_bwd", "sigmoid_dice_fwd", "sigmoid_loss_bwd", "adam_step", "pack_weights",
          "conv3d_pack_up_weights", "pack_weights_batched"):
    wrap(n)
B, X, Y, C = 64, 256, 256, 5
eng = UNetEngine(UNetPlan(C, (X, Y), depth=4, n_base_filters=32, ndim=2), B, dtype=torch.bfloat16)
eng._wg_stream = None
g = torch.Generator().manual_seed(0)
x = torch.randn((1, B, X, Y, C), generator=g).cuda().to(torch.bfloat16)
y = (torch.rand((B * X * Y,), generator=g) > 0.7).to(torch.uint8).cuda()
for _ in range(3):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
on[0] = True
K = 5
for _ in range(K):
    eng.train_step(x, y, 1e-4)
torch.cuda.synchronize()
PEAK = 2500.0          # dense bf16 TFLOP/s (MI355X guide)


def conv_flops(name, s):
    """algorithmic flops of one launch by the reference's tap count (3 x 3 taps per output pixel): 2 * 9 * Cin * Cout * pixels.
    s = shapes of the first seven positional arguments (None where the argument is no tensor); planar tensors are [1][slices][H][W][C]"""
    def pix(t):
        return t[0] * t[1] * t[2] * t[3]

    def ch(t):
        return 0 if t is None else t[4]
    if name == "conv3d_fwd":                    # (src0, src1, w, bias, y)
        return 2.0 * 9 * (ch(s[0]) + ch(s[1])) * ch(s[4]) * pix(s[4])
    if name == "conv3d_fwd_tail":               # (src0, w, bias, y)
        return 2.0 * 9 * ch(s[0]) * ch(s[3]) * pix(s[3])
    if name == "conv3d_dgrad":                  # (dy, w_dgrad, dx)
        return 2.0 * 9 * ch(s[0]) * ch(s[2]) * pix(s[0])
    if name == "conv3d_wgrad":                  # (src0, src1, dy, dw, db)
        return 2.0 * 9 * (ch(s[0]) + ch(s[1])) * ch(s[2]) * pix(s[2])
    if name == "conv3d_upcat_fwd":              # (src0_low, src1, w_up_f, w_sk_f, bias, y)
        return 2.0 * 9 * (ch(s[0]) + ch(s[1])) * ch(s[5]) * pix(s[5])
    if name == "conv3d_upcat_dgrad":            # (dy, w_up_d, w_sk_d, mask_low, mask_skip, dx_low, dx_skip)
        return 2.0 * 9 * (ch(s[5]) + ch(s[6])) * ch(s[0]) * pix(s[0])
    if name == "conv3d_upcat_wgrad":            # (src0_low, src1, dy, dw, db, dwc)
        return 2.0 * 9 * (ch(s[0]) + ch(s[1])) * ch(s[2]) * pix(s[2])
    return None


def conv_bytes(name, s):
    """algorithmic bf16 bytes of one launch: each tensor once (inputs + output + filters; the input-gradient form also reads the ReLU mask)"""
    def sz(t):
        return 0 if t is None else 2.0 * t[0] * t[1] * t[2] * t[3] * t[4]
    if name == "conv3d_fwd":
        return sz(s[0]) + sz(s[1]) + sz(s[4]) + 2.0 * 9 * ((s[0][4] if s[0] else 0) + (s[1][4] if s[1] else 0)) * s[4][4]
    if name == "conv3d_fwd_tail":
        return sz(s[0]) + sz(s[3]) + 2.0 * 9 * s[0][4] * s[3][4]
    if name == "conv3d_dgrad":
        return sz(s[0]) + 2 * sz(s[2]) + 2.0 * 9 * s[0][4] * s[2][4]
    if name == "conv3d_wgrad":
        return sz(s[0]) + sz(s[1]) + sz(s[2]) + 4.0 * 9 * ((s[0][4] if s[0] else 0) + (s[1][4] if s[1] else 0)) * s[2][4]
    if name == "conv3d_upcat_fwd":
        return sz(s[0]) + sz(s[1]) + sz(s[5]) + 2.0 * 9 * (s[0][4] + s[1][4]) * s[5][4]
    if name == "conv3d_upcat_dgrad":
        return sz(s[0]) + 2 * sz(s[5]) + 2 * sz(s[6]) + 2.0 * 9 * (s[5][4] + s[6][4]) * s[0][4]
    if name == "conv3d_upcat_wgrad":
        return sz(s[0]) + sz(s[1]) + sz(s[2]) + 4.0 * 9 * (s[0][4] + s[1][4]) * s[2][4]
    return None


PEAK_HBM = 8000.0      # GB/s (MI355X guide)
rows = []
for (name, shp), ev in rec.items():
    # median over the K steps: a pair of events also spans whatever the HOST does between a launch's kernels - one stalled step put 7.9 ms
    # on a 0.7 ms launch in the mean of the first r06 collection
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    ms = ts[len(ts) // 2]
    n = len(ev) // K
    fl = conv_flops(name, shp)
    by = conv_bytes(name, shp) if fl else None
    # the launch's own roofline: an MFMA-bound time and an HBM-bound time, whichever is longer (3 x 3 convs have a third of the 3-D layers'
    # arithmetic intensity: the full-resolution 2-D layers are HBM-bound)
    t_mfma = fl / (PEAK * 1e12) * 1e3 if fl else None
    t_hbm = by / (PEAK_HBM * 1e9) * 1e3 if by else None
    rows.append({"op": name, "shapes": [list(t) for t in shp if t is not None], "launches_per_step": n, "ms_per_launch": round(ms, 4), "ms_per_step": round(ms * n, 4),
                 "gflop": round(fl / 1e9, 1) if fl else None, "mfma_frac": round(fl / (ms * 1e-3) / 1e12 / PEAK, 4) if fl else None,
                 "algorithmic_mb": round(by / 1e6, 1) if by else None, "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM, 4) if by else None,
                 "roofline_ms": round(max(t_mfma, t_hbm), 4) if fl else None, "bound": ("hbm" if t_hbm > t_mfma else "mfma") if fl else None,
                 "roofline_frac": round(max(t_mfma, t_hbm) / ms, 4) if fl else None})
rows.sort(key=lambda r: -r["ms_per_step"])
tot = sum(r["ms_per_step"] for r in rows)
conv = [r for r in rows if r["gflop"]]
cms, cfl = sum(r["ms_per_step"] for r in conv), sum(r["gflop"] * r["launches_per_step"] for r in conv)
croof = sum(r["roofline_ms"] * r["launches_per_step"] for r in conv)
for r in rows:
    print("%8.3f ms  x%d  %-22s %-16s %s" % (r["ms_per_step"], r["launches_per_step"], r["op"],
                                            "" if r["mfma_frac"] is None else "%.2f mfma %.2f %s" % (r["mfma_frac"], r["roofline_frac"], r["bound"]), r["shapes"]))
print("total %.3f ms per step (exclusive); 3x3 convs %.3f ms, %.0f GFLOP algorithmic = %.3f of the MFMA peak; sum of the launches' own rooflines "
      "(max of MFMA- and HBM-bound time) %.3f ms = %.3f of the measured" % (tot, cms, cfl, cfl / cms / PEAK if cms else 0, croof, croof / cms if cms else 0))
out = {"workload": "configs[3]: 2-D U-Net depth 4 / 32 filters, batch %d x %dx%dx%d, bf16, one stream (exclusive HIP-event times, median over %d steps)" % (B, X, Y, C, K),
       "rows": rows, "total_ms_per_step": round(tot, 3), "conv_ms_per_step": round(cms, 3), "conv_gflop_per_step": round(cfl, 1),
       "conv_mfma_frac": round(cfl / cms / PEAK, 4) if cms else None, "conv_roofline_ms_per_step": round(croof, 3),
       "conv_roofline_frac": round(croof / cms, 4) if cms else None,
       "note": "roofline_ms of a launch = max(algorithmic FLOPs / 2.5 PF, algorithmic bf16 bytes / 8 TB/s): the full-resolution 3 x 3 layers are HBM-bound"}
if len(sys.argv) > 1:
    import bench
    out["kernel_source_hash"] = bench.kernel_source_hash()
    out["git_head"] = sys.argv[2] if len(sys.argv) > 2 else None
    json.dump(out, open(sys.argv[1], "w"), indent=1)
