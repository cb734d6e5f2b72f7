"""bench.py's N-rank launcher (driver contract: `python bench.py --gpus N` must produce an N-rank line; SURVEY 8e, north_star
"patches/sec at 1/2/4/8 GPUs").  Runs on the CPU: `--selftest-cpu` swaps the engine for a gloo all-reduce, everything else -
argument handling, child start before any GPU call, rendezvous on 127.0.0.1, relay of rank 0's line, exit codes - is the real path."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def _run(args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env or _clean_env(), timeout=timeout)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return json.loads(lines[0])


def test_bare_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--selftest-cpu", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["n_gpus"] == 2 and d["collective_ranks"] == 2
    assert d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"


def test_single_rank_line():
    d = _line(_run(["--selftest-cpu", "--steps", "2", "--warmup", "0"]))
    assert d["n_gpus"] == 1 and d["collective_ranks"] == 1


def test_under_an_external_launcher_each_process_is_one_rank():
    """the driver's form: torch.distributed.run sets RANK / WORLD_SIZE / MASTER_* and starts bench.py --gpus N once per rank"""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--selftest-cpu", "--steps", "2", "--warmup", "0"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])["n_gpus"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]          # only rank 0 prints


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = _run(["--gpus", "2", "--selftest-cpu"], env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    assert not r.stdout.strip()


def test_more_gpus_than_devices_is_refused_without_touching_the_gpu():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(n, 2))])
    assert r.returncode == 2 and "visible" in r.stderr
