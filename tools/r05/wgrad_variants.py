#!/usr/bin/env python3
"""kd-sharing weight-gradient launch of the three big plain layers under several builds of the library (FMRI_LIB), interleaved rounds, one
child process per (round, library); prints min and median ms per layer.   usage: wgrad_variants.py lib1.so lib2.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAYERS = (("dec0b", 64, 64, 64, 128, 128), ("enc0b", 32, 64, 64, 128, 128), ("dec1b", 128, 128, 32, 64, 64))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
    import torch
    from fmri_hip import ops
    out = {}
    for name, C0, Cout, D, H, W in LAYERS:
        N = 4
        x = torch.randn((N, D, H, W, C0), device="cuda").to(torch.bfloat16)
        dy = torch.randn((N, D, H, W, Cout), device="cuda").to(torch.bfloat16)
        dw = torch.zeros((27, Cout, C0), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        for _ in range(200):
            ops.conv3d_wgrad(x, None, dy, dw, db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ops.conv3d_wgrad(x, None, dy, dw, db)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 100
    print("RESULT " + json.dumps(out))
    sys.exit(0)
libs = sys.argv[1:]
acc = {l: [] for l in libs}
for rd in range(4):
    for l in libs:
        env = dict(os.environ, FMRI_LIB=os.path.abspath(l))
        o = subprocess.check_output([sys.executable, __file__, "--child"], env=env).decode()
        acc[l].append(json.loads([x for x in o.splitlines() if x.startswith("RESULT ")][0][7:]))
for l in libs:
    print("%-28s" % os.path.basename(l), "  ".join("%s min %.4f med %.4f" % (n[0], min(r[n[0]] for r in acc[l]), sorted(r[n[0]] for r in acc[l])[2]) for n in LAYERS))
