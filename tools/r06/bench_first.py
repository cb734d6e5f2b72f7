#!/usr/bin/env python3
"""First-layer kernels (`k_conv_first_fwd / _wgrad`, csrc/conv3d_first.hip) at the shapes of configs[1] (1 -> 32 @ 4 x 64x128x128) and
configs[3] (5 slices -> 32 @ 64 x 256x256, planar): time per launch and a digest of the results, per library, interleaved - arms with
equal digests computed the same bits (the weight gradient is summed with atomics: its digest is of the result rounded to 2^-6 on dyadic
data, which is exact).  usage: bench_first.py --libs a.so,b.so [--iters 20]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))

CASES = [("cfg1_3d", 1, 32, 4, 64, 128, 128, False), ("cfg3_2d", 5, 32, 1, 64, 256, 256, True)]


def run(iters):
    import torch
    from fmri_hip import ops
    res = {}
    g = torch.Generator(device="cpu").manual_seed(7)
    for name, C0, Cout, N, D, H, W, planar in CASES:
        # dyadic data: every product and partial sum is exact in fp32, so the results do not depend on the order of summation
        x = (torch.randint(-8, 9, (N, D, H, W, C0), generator=g).float() / 8).to(torch.bfloat16).cuda()
        w = (torch.randint(-4, 5, (27, Cout, C0), generator=g).float() / 16).to(torch.bfloat16).cuda()
        b = (torch.randint(-4, 5, (Cout,), generator=g).float() / 8).cuda()
        dy = (torch.randint(-2, 3, (N, D, H, W, Cout), generator=g).float() / 4).to(torch.bfloat16).cuda()
        y = torch.empty((N, D, H, W, Cout), device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros((27, Cout, C0), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        for kind in ("fwd", "wgrad"):
            f = (lambda: ops.conv3d_fwd(x, None, w, b, y, planar=planar)) if kind == "fwd" else \
                (lambda: ops.conv3d_wgrad(x, None, dy, dw, db, planar=planar))
            dw.zero_(); db.zero_()
            f()
            torch.cuda.synchronize()
            t = y if kind == "fwd" else torch.cat([dw.reshape(-1), db])
            digest = hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:12]
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res["%s_%s" % (name, kind)] = (e0.elapsed_time(e1) / iters, digest, N * D * H * W * Cout * 2 / 1e6)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or not a.libs:
        r = run(a.iters)
        print("RESULT " + json.dumps(r))
        sys.exit(0)
    libs = a.libs.split(",")
    acc = {l: [] for l in libs}
    for rd in range(3):
        for l in libs:
            path, _, kv = l.partition("@")          # "lib.so@VAR=VAL" adds an environment switch to that arm
            env = dict(os.environ, FMRI_LIB=os.path.abspath(path))
            if kv:
                env[kv.split("=")[0]] = kv.split("=")[1]
            out = subprocess.check_output([sys.executable, __file__, "--child", "--iters", str(a.iters)], env=env).decode()
            acc[l].append(json.loads([x for x in out.splitlines() if x.startswith("RESULT ")][0][7:]))
    print("%-14s " % "launch" + " ".join("%34s" % os.path.basename(l)[-30:].replace("libfmri_hip_", "") for l in libs))
    for k in acc[libs[0]][0]:
        row = []
        for l in libs:
            ms = min(r[k][0] for r in acc[l])
            row.append("%7.1f us %5.2f TB/s %s" % (ms * 1e3, acc[l][0][k][2] / ms / 1e3, acc[l][0][k][1]))
        print("%-14s " % k + " ".join("%34s" % x for x in row))
