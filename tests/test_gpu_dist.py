"""The data-parallel context on real RCCL: a 1-rank `nccl` process group on the one GPU of the box, collectives forced on, so that the
bucketed gradient all-reduce on the side stream and the Dice-sum all-reduce really go through RCCL (identity reductions: the
trajectory must equal the plain engine's up to atomics ordering).  The multi-rank logic itself is covered on CPU by tests/test_dist_gloo.py."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "fetal-mri-segmentation_amd"))
    import numpy as np, torch, torch.distributed as dist
    from fmri_hip.engine import UNetEngine, UNetPlan
    from fmri_hip.dist import DataParallel
    from oracle import unet_oracle as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    sp = (16, 32, 32)
    x, y = O.synthetic_batch((2, 1) + sp)
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16).reshape(2, *sp, 1).contiguous()
    yd = torch.from_numpy(y).cuda().reshape(-1).contiguous()
    out = []
    for use in (False, True):
        ctx = DataParallel(1, 0, bucket_bytes=1 << 20, force_collectives=True) if use else None
        eng = UNetEngine(UNetPlan(1, sp, depth=3, n_base_filters=32), 2, dtype=torch.bfloat16, dist_ctx=ctx, seed=3)
        sums = [eng.train_step(xd, yd, 1e-3).cpu().numpy().copy() for _ in range(3)]
        torch.cuda.synchronize()
        out.append((np.stack(sums), eng.P.cpu().numpy().copy(), [] if ctx is None else list(ctx.launched)))
    assert len(out[1][2]) >= 2, out[1][2]                      # several buckets went through RCCL
    assert out[1][2][-1][1] == eng.n_flat and out[1][2][0][0] == 0
    # identity reductions: same trajectory up to the summation order of the fp32 atomics in the weight-gradient kernels
    assert np.allclose(out[0][0], out[1][0], rtol=2e-3, atol=1e-3), (out[0][0], out[1][0])
    assert float(np.abs(out[0][1] - out[1][1]).mean()) <= 2e-4
    dist.destroy_process_group()
    print("RCCL_OK buckets", len(out[1][2]))
""") % (ROOT, ROOT)


def test_rccl_single_rank_bucketed_allreduce(tmp_path):
    f = tmp_path / "rccl1.py"
    f.write_text(SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
