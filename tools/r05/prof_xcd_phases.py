#!/usr/bin/env python3
"""Per-XCD split of the forward kernel's phase profile (instrumented build): which part of a phase is longer on the XCDs that finish late?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_conv import LAYERS
from fmri_hip import ops, _lib
L = _lib.lib()
L.fmri_debug_prof_xcd.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 128)()
N = 4
for name, C0, up0, C1, Cout, D, H, W in LAYERS:
    if up0 or name not in ("enc0b", "dec0b", "dec1b", "enc1b"):
        continue
    src0 = torch.randn((N, D, H, W, C0), device="cuda").to(torch.bfloat16)
    w = (torch.randn((27, Cout, C0), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
    for _ in range(300):
        ops.conv3d_fwd(src0, None, w, b, y)
    torch.cuda.synchronize()
    L.fmri_debug_prof_xcd(None, 1)
    for _ in range(20):
        ops.conv3d_fwd(src0, None, w, b, y)
    torch.cuda.synchronize()
    L.fmri_debug_prof_xcd(buf, 0)
    q = list(buf)
    print(name)
    for x in range(8):
        p = q[x * 16:x * 16 + 16]
        ph = max(p[6], 1)
        print("  xcd %d: consumer cycles/phase %6.0f (barrier %5.0f, mfma-loop %5.0f, epilogue %5.0f) | producer per phase: dma-wait %5.0f barrier %5.0f issue %5.0f" % (
            x, p[5] / ph, p[1] / ph, p[3] / ph, p[4] / ph, p[12] / ph, p[13] / ph, p[14] / ph))
