#!/usr/bin/env python3
"""How far apart do the XCDs finish a persistent forward-type launch?  Needs the instrumented library (make prof); per workgroup the kernel
records s_memrealtime at entry / exit and its XCC_ID.  Prints, per layer, the launch span and per XCD the mean exit time relative to it."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_conv import LAYERS
from fmri_hip import ops, _lib
L = _lib.lib()
L.fmri_debug_prof_wg.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * (3 * 1024))()
N = 4
for name, C0, up0, C1, Cout, D, H, W in LAYERS:
    if up0 or name not in ("enc0b", "dec0b", "dec1b", "enc1b"):
        continue
    src0 = torch.randn((N, D, H, W, C0), device="cuda").to(torch.bfloat16)
    w = (torch.randn((27, Cout, C0), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
    for _ in range(300):                      # ~0.2 s of back-to-back launches: the clocks of this load
        ops.conv3d_fwd(src0, None, w, b, y)
    torch.cuda.synchronize()
    L.fmri_debug_prof_wg(buf)
    q = list(buf)
    wgs = [(q[3 * i], q[3 * i + 1], int(q[3 * i + 2])) for i in range(256) if q[3 * i + 1]]
    t0, t1 = min(a for a, _, _ in wgs), max(e for _, e, _ in wgs)
    span = (t1 - t0) / 100.0            # us
    per = {}
    for a, e, x in wgs:
        per.setdefault(x, []).append((e - t0) / 100.0)
    print("%-6s launch span %.1f us | per XCD (workgroups: mean exit / last exit, %% of the span): %s" % (
        name, span, "  ".join("%d: %d wg %.1f / %.1f" % (x, len(v), 100 * sum(v) / len(v) / span, 100 * max(v) / span) for x, v in sorted(per.items()))))
    dur = {}
    for a, e, x in wgs:
        dur.setdefault(x, []).append(((a - t0) / 100.0, (e - a) / 100.0))
    print("       per XCD mean entry us / mean duration us: %s" % "  ".join("%d: %.1f / %.1f" % (x, sum(v[0] for v in d) / len(d), sum(v[1] for v in d) / len(d)) for x, d in sorted(dur.items())))
    idle = sum(span - (e - t0) / 100.0 for _, e, _ in wgs) / len(wgs)
    print("       mean idle time of a CU behind its workgroup's exit: %.1f us = %.1f %% of the launch" % (idle, 100 * idle / span))
