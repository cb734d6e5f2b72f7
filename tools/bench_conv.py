#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at BASELINE config-2 shapes (tuning aid; prints TFLOP/s per layer).
Usage: python tools/bench_conv.py [--libs a.so,b.so] [--iters 20]   (several libs = interleaved A/B in ONE process via
subprocesses pinned to the same device)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))

LAYERS = [  # name, C0, up0, C1, Cout, D,H,W   (N = 4)
    ("enc0b", 32, 0, 0, 64, 64, 128, 128), ("enc1a", 64, 0, 0, 64, 32, 64, 64), ("enc1b", 64, 0, 0, 128, 32, 64, 64),
    ("enc2a", 128, 0, 0, 128, 16, 32, 32), ("enc2b", 128, 0, 0, 256, 16, 32, 32), ("enc3a", 256, 0, 0, 256, 8, 16, 16),
    ("enc3b", 256, 0, 0, 512, 8, 16, 16), ("dec2a", 512, 1, 256, 256, 16, 32, 32), ("dec2b", 256, 0, 0, 256, 16, 32, 32),
    ("dec1a", 256, 1, 128, 128, 32, 64, 64), ("dec1b", 128, 0, 0, 128, 32, 64, 64), ("dec0a", 128, 1, 64, 64, 64, 128, 128),
    ("dec0b", 64, 0, 0, 64, 64, 128, 128),
]


def run(iters, which):
    import torch
    from fmri_hip import ops
    N = 4
    res = {}
    for name, C0, up0, C1, Cout, D, H, W in LAYERS:
        s0 = (N, D // 2, H // 2, W // 2, C0) if up0 else (N, D, H, W, C0)
        src0 = torch.randn(s0, device="cuda").to(torch.bfloat16)
        src1 = torch.randn((N, D, H, W, C1), device="cuda").to(torch.bfloat16) if C1 else None
        w = (torch.randn((27, Cout, C0 + C1), device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.zeros(Cout, device="cuda")
        y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros((27, Cout, C0 + C1), device="cuda")
        nws = ops.conv3d_wgrad_workspace_bytes(C0, C1, Cout, N, D, H, W, torch.bfloat16) if os.environ.get("BENCH_WGRAD_WS", "1") == "1" else 0
        ws = torch.empty(max(nws // 4, 1), device="cuda") if nws else None
        db = torch.zeros(Cout, device="cuda")
        fl = 2.0 * 27 * (C0 + C1) * Cout * N * D * H * W
        for kind in which:
            f = (lambda: ops.conv3d_fwd(src0, src1, w, b, y, up0=bool(up0))) if kind == "fwd" else \
                (lambda: ops.conv3d_wgrad(src0, src1, y, dw, db, up0=bool(up0), workspace=ws))
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            res["%s_%s" % (name, kind)] = (ms, fl / ms / 1e9)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--which", default="fwd,wgrad")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or not a.libs:
        r = run(a.iters, a.which.split(","))
        if a.child:
            print("RESULT " + json.dumps(r))
        else:
            tot = {}
            for k, (ms, tf) in r.items():
                print("%-14s %8.3f ms %8.1f TFLOP/s" % (k, ms, tf))
                tot[k.split("_")[1]] = tot.get(k.split("_")[1], 0) + ms
            print(tot)
        sys.exit(0)
    libs = a.libs.split(",")
    rounds = 3
    acc = {l: [] for l in libs}
    for rd in range(rounds):
        for l in libs:
            path, _, kv = l.partition("@")          # "lib.so@VAR=VAL" adds an environment switch to that arm
            env = dict(os.environ, FMRI_LIB=os.path.abspath(path))
            if kv:
                env[kv.split("=")[0]] = kv.split("=")[1]
            out = subprocess.check_output([sys.executable, __file__, "--child", "--iters", str(a.iters), "--which", a.which], env=env).decode()
            acc[l].append(json.loads([x for x in out.splitlines() if x.startswith("RESULT ")][0][7:]))
    keys = list(acc[libs[0]][0].keys())
    print("%-14s " % "layer" + " ".join("%22s" % os.path.basename(l)[-22:] for l in libs))
    tot = {l: 0.0 for l in libs}
    for k in keys:
        row = []
        for l in libs:
            ms = min(r[k][0] for r in acc[l])
            tot[l] += ms
            row.append("%9.3f ms %7.0f TF" % (ms, acc[l][0][k][1] * acc[l][0][k][0] / ms))
        print("%-14s " % k + " ".join("%22s" % x for x in row))
    print("%-14s " % "TOTAL(min)" + " ".join("%19.3f ms" % tot[l] for l in libs))
