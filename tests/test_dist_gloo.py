"""N > 1 path on CPU: two gloo ranks drive the bucketed gradient all-reduce and the global-Dice sum exchange of
fmri_hip.dist.DataParallel with a stand-in engine (flat gradient buffer in backward-completion order)."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeEngine:
    def __init__(self, sizes):
        self.layout = OrderedDict()
        off = 0
        for i, (nw, nb) in enumerate(sizes):
            self.layout["l%d" % i] = dict(w=(off, nw), b=(off + nw, nb))
            off += nw + nb
        self.n_flat = off
        self.G = torch.zeros(off)
        self.P = torch.zeros(off)

    def refresh_weight_copies(self):
        pass


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmri_hip.dist import DataParallel
    eng = FakeEngine([(1000, 10), (5000, 20), (300, 5), (70000, 30), (10, 1)])
    dp = DataParallel(world, rank, bucket_bytes=4 * 6000)
    # parameters broadcast from rank 0
    eng.P.fill_(float(rank + 1))
    dp.broadcast_params(eng)
    assert float(eng.P[0]) == 1.0
    # backward: gradients become ready layer by layer
    g = torch.Generator().manual_seed(100 + rank)
    full = torch.randn(eng.n_flat, generator=g)
    for name, L in eng.layout.items():
        lo, hi = L["w"][0], L["b"][0] + L["b"][1]
        eng.G[lo:hi] = full[lo:hi]
        dp.grad_ready(eng, name)
    dp.finish(eng)
    expect = sum(torch.randn(eng.n_flat, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    ok = bool(torch.allclose(eng.G, expect, atol=1e-6))
    covered = sorted(dp.launched)
    contiguous = covered[0][0] == 0 and covered[-1][1] == eng.n_flat and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    sums = torch.tensor([1.0 + rank, 2.0, 3.0, 0, 0, 0, 0, 8.0], dtype=torch.float64)
    dp.all_reduce_sums(sums)
    q.put((rank, ok, contiguous, len(covered), sums.tolist(), dp.grad_scale))
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, ok, contiguous, nb, sums, gs in res:
        assert ok and contiguous, (rank, ok, contiguous)
        assert nb >= 2                                   # more than one bucket was launched
        assert sums[0] == 3.0 and sums[1] == 4.0 and sums[7] == 16.0
        assert gs == 1.0                                 # exact global-batch Dice -> gradients are summed
