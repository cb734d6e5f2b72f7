#!/usr/bin/env python3
"""Where a fwd-kernel phase spends its cycles (tuning aid).  Needs the instrumented library (make -C fetal-mri-segmentation_amd/csrc prof):
   FMRI_LIB=.../libfmri_hip_prof.so python tools/prof_phases.py
Prints, per layer, the share of wave time in: DMA wait | barrier wait | DMA issue | fragment reads + MFMA issue | rest."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_conv import LAYERS
from fmri_hip import ops, _lib

L = _lib.lib()
L.fmri_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 12)()
try:
    L.fmri_debug_prof_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
    pbuf = (ctypes.c_ulonglong * 36)()
except AttributeError:
    pbuf = None
N = 4
for name, C0, up0, C1, Cout, D, H, W in LAYERS:
    s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
    src0 = torch.randn(s0, device="cuda").to(torch.bfloat16)
    src1 = torch.randn((N, D, H, W, C1), device="cuda").to(torch.bfloat16) if C1 else None
    w = (torch.randn((27, Cout, C0 + C1), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv3d_fwd(src0, src1, w, b, y, up0=bool(up0))
    torch.cuda.synchronize()
    L.fmri_debug_prof(None, 1)
    if pbuf is not None:
        L.fmri_debug_prof_phases(None, 1)
    ops.conv3d_fwd(src0, src1, w, b, y, up0=bool(up0))
    torch.cuda.synchronize()
    L.fmri_debug_prof(buf, 0)
    if pbuf is not None:
        L.fmri_debug_prof_phases(pbuf, 0)
        q = list(pbuf)
        for lab, o in (("drain items", 0), ("other items", 9)):
            if sum(q[18 + o:27 + o]):
                print("%-7s producers' issue cycles per phase index, %s: %s" % (name, lab, " ".join("%5.0f" % (q[o + i] / max(q[18 + o + i], 1)) for i in range(9))))
    p = list(buf)
    tot = max(p[5], 1)
    sec = [p[i] / tot * 100 for i in range(4)]
    ep = p[4] / tot * 100
    if p[10]:      # warp-specialised kernel: barrier 1 | pack + stage | barrier 2 | read back + store
        epd = "(barrier %.1f, pack+stage %.1f, barrier %.1f, store %.1f)" % tuple(p[i] / tot * 100 for i in (7, 8, 9, 10))
    else:
        epd = "(barrier %.1f, bias+pack+lds-write %.1f, lds-read+store %.1f)" % (p[7] / tot * 100, p[8] / tot * 100, (p[4] - p[7] - p[8]) / tot * 100)
    if p[11]:     # warp-specialised kernel: the PRODUCER waves' phase (same clock, same number of waves): counted DMA wait | barrier wait | issue
        print("%-7s producers: dma-wait %5.1f%%  barrier %5.1f%%  issue (filter slab + drain part + halo pieces) %5.1f%%" % (
            name, p[0] / tot * 100, p[11] / tot * 100, p[2] / tot * 100))
        sec[0] = sec[2] = 0.0
    print("%-7s cyc/phase %6.0f | dma-wait %5.1f%%  barrier %5.1f%%  dma-issue %5.1f%%  mfma-loop %5.1f%%  epilogue %5.1f%% %s  other %5.1f%%"
          % (name, tot / max(p[6], 1), sec[0], sec[1], sec[2], sec[3], ep, epd, 100 - sum(sec) - ep))


def report(name):
    L.fmri_debug_prof(buf, 0)
    p = list(buf)
    tot = max(p[5], 1)
    if pbuf is not None:
        L.fmri_debug_prof_phases(pbuf, 0)
        q = list(pbuf)
        for lab, o in (("drain items", 0), ("other items", 9)):
            if sum(q[18 + o:27 + o]):
                print("%-12s producers' issue cycles per phase index, %s: %s" % (name, lab, " ".join("%5.0f" % (q[o + i] / max(q[18 + o + i], 1)) for i in range(9))))
    print("%-12s cyc/phase %6.0f | producers: dma-wait %4.1f%% barrier %4.1f%% issue %4.1f%% | consumers: barrier %4.1f%% mfma-loop %4.1f%% epilogue %4.1f%%" % (
        name, tot / max(p[6], 1), p[0] / tot * 100, p[11] / tot * 100, p[2] / tot * 100, p[1] / tot * 100, p[3] / tot * 100, p[4] / tot * 100))


def reset():
    L.fmri_debug_prof(None, 1)
    if pbuf is not None:
        L.fmri_debug_prof_phases(None, 1)


if "--more" in sys.argv:
    # the launches prof above does not reach: input gradients with a ReLU mask (NT = 1 for enc0b) and the parity form of the decoder 'a' layers
    bf = torch.bfloat16
    for name, Cin, Cout, D, H, W in (("enc0b dgrad", 32, 64, 64, 128, 128), ("dec0b dgrad", 64, 64, 64, 128, 128), ("enc1b dgrad", 64, 128, 32, 64, 64)):
        dy = torch.randn((N, D, H, W, Cout), device="cuda").to(bf)
        wd = (torch.randn((27, Cin, Cout), device="cuda") * 0.05).to(bf)
        dx = torch.empty((N, D, H, W, Cin), device="cuda", dtype=bf)
        mask = torch.randn((N, D, H, W, Cin), device="cuda").to(bf)
        for _ in range(3):
            ops.conv3d_dgrad(dy, wd, dx, mask=mask)
        torch.cuda.synchronize()
        reset()
        ops.conv3d_dgrad(dy, wd, dx, mask=mask)
        torch.cuda.synchronize()
        report(name)
    for name, C0, C1, Cout, D, H, W in (("dec2a", 512, 256, 256, 16, 32, 32), ("dec1a", 256, 128, 128, 32, 64, 64), ("dec0a", 128, 64, 64, 64, 128, 128)):
        xl = torch.randn((N, D // 2, H // 2, W // 2, C0), device="cuda").to(bf)
        xs = torch.randn((N, D, H, W, C1), device="cuda").to(bf)
        w = torch.randn((27, Cout, C0 + C1), device="cuda") * 0.05
        b = torch.zeros(Cout, device="cuda")
        y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=bf)
        up_f, up_d = torch.empty((8, 8, Cout, C0), device="cuda", dtype=bf), torch.empty((8, 8, C0, Cout), device="cuda", dtype=bf)
        sk_f, sk_d = torch.empty((27, Cout, C1), device="cuda", dtype=bf), torch.empty((27, C1, Cout), device="cuda", dtype=bf)
        ops.conv3d_pack_up_weights(w, C0, C1, up_f, up_d, sk_f, sk_d)
        dy = torch.randn((N, D, H, W, Cout), device="cuda").to(bf)
        dxl, dxs = torch.empty_like(xl), torch.empty_like(xs)
        for what, f in (("upcat fwd (MODE 1 + skip launch)", lambda: ops.conv3d_upcat_fwd(xl, xs, up_f, sk_f, b, y)),
                        ("upcat dgrad (MODE 2 + skip launch)", lambda: ops.conv3d_upcat_dgrad(dy, up_d, sk_d, xl, None, dxl, dxs))):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            reset()
            f()
            torch.cuda.synchronize()
            report(name + " " + what[:11])
