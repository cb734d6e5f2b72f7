#!/usr/bin/env python3
"""k_sigmoid_dice_fwd (csrc/pointwise.hip) at the size of the configs[1] step (4 x 64x128x128 logits): time per launch and the ten sums,
per library / environment arm.  usage: bench_dice.py --libs a.so,b.so@FMRI_DICE_WG=512"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))


def run(iters):
    import torch
    from fmri_hip import ops
    n = 4 * 64 * 128 * 128
    g = torch.Generator(device="cpu").manual_seed(3)
    logits = (torch.randn(n, generator=g) * 3).cuda()
    y = (torch.rand(n, generator=g) < 0.2).to(torch.uint8).cuda()
    probs = torch.empty(n, device="cuda")
    sums = torch.zeros(16, device="cuda", dtype=torch.float64)
    ops.sigmoid_dice_fwd(logits, y, probs, sums)
    torch.cuda.synchronize()
    first = sums.cpu().tolist()[:10]
    psum = float(probs.double().sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ops.sigmoid_dice_fwd(logits, y, probs, sums)
    e0.record()
    for _ in range(iters):
        ops.sigmoid_dice_fwd(logits, y, probs, sums)
    e1.record()
    torch.cuda.synchronize()
    return {"us": e0.elapsed_time(e1) / iters * 1e3, "sums": first, "probs_sum": psum}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or not a.libs:
        print("RESULT " + json.dumps(run(a.iters)))
        sys.exit(0)
    for rd in range(2):
        for l in a.libs.split(","):
            path, _, kv = l.partition("@")
            env = dict(os.environ, FMRI_LIB=os.path.abspath(path))
            if kv:
                env[kv.split("=")[0]] = kv.split("=")[1]
            out = subprocess.check_output([sys.executable, __file__, "--child", "--iters", str(a.iters)], env=env).decode()
            r = json.loads([x for x in out.splitlines() if x.startswith("RESULT ")][0][7:])
            print("%-44s %7.1f us  sums %s  probs %.6f" % (os.path.basename(l).replace("libfmri_hip_", ""), r["us"], " ".join("%.9g" % v for v in r["sums"]), r["probs_sum"]))
