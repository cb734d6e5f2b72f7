#!/usr/bin/env python3
"""Where a weight-gradient workgroup spends its time per unit (tuning aid; needs `make -C fetal-mri-segmentation_amd/csrc prof`):
   FMRI_LIB=.../libfmri_hip_prof.so python tools/prof_wgrad.py
DMA wait | barrier | DMA issue | fragment reads + MFMAs | rest (flush, prologue), in s_memtime ticks per unit (72 MFMAs per wave = 2304 pipe cycles)."""
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
L.fmri_debug_prof_wgrad.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 12)()
N = 4
UP = "--upcat" in sys.argv          # the parity-form weight gradient of the decoder 'a' layers (k_conv_wgrad_up_kd / k_conv_wgrad_mfma<.., UPW>)
for name, C0, up0, C1, Cout, D, H, W in LAYERS:
    if UP != bool(up0):
        continue
    if UP:
        xl = torch.randn((N, D // 2, H // 2, W // 2, C0), device="cuda").to(torch.bfloat16)
        dy = torch.randn((N, D, H, W, Cout), device="cuda").to(torch.bfloat16)
        dw = torch.zeros((27, Cout, C0), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        scratch = torch.empty(64 * Cout * C0, device="cuda")
        for _ in range(2):
            ops.conv3d_upcat_wgrad(xl, None, dy, dw, db, scratch)
        torch.cuda.synchronize()
        L.fmri_debug_prof_wgrad(None, 1)
        ops.conv3d_upcat_wgrad(xl, None, dy, dw, db, scratch)
        torch.cuda.synchronize()
        L.fmri_debug_prof_wgrad(buf, 0)
        p = list(buf)
        tot, units = max(p[5], 1), max(p[6], 1)
        sec = [p[i] / tot * 100 for i in range(4)]
        print("%-7s ticks/unit %6.0f | dma-wait %5.1f%%  barrier %5.1f%%  set-up %5.1f%%  reads+mfma+dma-issue %5.1f%% (%5.0f ticks/unit; 64 MFMAs per wave)  fresh-column tail %4.1f%%  rest %5.1f%%"
              % (name, tot / units, sec[0], sec[1], sec[2], sec[3], p[3] / units, p[4] / tot * 100, 100 - sum(sec) - p[4] / tot * 100))
        continue
    x = torch.randn((N, D, H, W, C0), device="cuda").to(torch.bfloat16)
    dy = torch.randn((N, D, H, W, Cout), device="cuda").to(torch.bfloat16)
    dw = torch.zeros((27, Cout, C0), device="cuda")
    db = torch.zeros(Cout, device="cuda")
    for _ in range(2):
        ops.conv3d_wgrad(x, None, dy, dw, db)
    torch.cuda.synchronize()
    L.fmri_debug_prof_wgrad(None, 1)
    ops.conv3d_wgrad(x, None, dy, dw, db)
    torch.cuda.synchronize()
    L.fmri_debug_prof_wgrad(buf, 0)
    p = list(buf)
    tot, units = max(p[5], 1), max(p[6], 1)
    sec = [p[i] / tot * 100 for i in range(4)]
    print("%-7s ticks/unit %6.0f | dma-wait %5.1f%%  barrier %5.1f%%  dma-issue %5.1f%%  reads+mfma %5.1f%% (%5.0f ticks/unit)  fresh-column tail %4.1f%%  rest (prologue) %5.1f%%"
          % (name, tot / units, sec[0], sec[1], sec[2], sec[3], p[3] / units, p[4] / tot * 100, 100 - sum(sec) - p[4] / tot * 100))
