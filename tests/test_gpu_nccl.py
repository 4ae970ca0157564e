"""The multi-GPU path on RCCL, on the ONE GPU a test box has.

A 1-rank `nccl` process group with `LoglikeExchange(force_collective=True)`: the real C2 and C5 steps of bench.py as
`GraphedStep`s, bench.py's own timed loop (`make_runner`: replay -> staging copy -> asynchronous collective from the graph's
static output -> consumer one step behind) for >= 50 steps with the leaves CHANGED between steps, and every exchanged
vector must equal -- bit for bit -- the per-draw scalars the same graph gives without any process group.  Each case runs in
a subprocess of its own (a process group in the pytest process would change what `shard_bounds` sees in other tests).
Also: `bench.py` started through torch.distributed.run (the driver's N > 1 form, with one rank) and the launcher command
`python bench.py --gpus N` turns itself into.

reference counterpart: none (docs/user/multiprocessing.rst:6-8 is one process per chain); SURVEY.md section 8(e).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, {root!r})
cfg, D, steps = {cfg!r}, {D}, {steps}
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "{port}")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
import bench
import exoplanet_amd as xo
from exoplanet_amd import ops
from exoplanet_amd.distributed import LoglikeExchange

wl = bench.WORKLOADS[cfg](xo, ops, dev, D, 0)
graph = xo.GraphedStep(wl.fn, *wl.leaves)
si = wl.scalar_index
base = [x.detach().clone() for x in wl.leaves]
k_r = wl.names.index("r")


def set_leaves(i):
    # a different planet radius every step: every step's scalars differ from the last step's
    with torch.no_grad():
        wl.leaves[k_r].copy_(base[k_r] * (1.0 + 1e-3 * i))


# 1. no process group: the scalars of every step
want = []
for i in range(steps):
    set_leaves(i)
    want.append(graph()[si].detach().clone())
torch.cuda.synchronize()
assert not dist.is_initialized()

# 2. the same steps through bench.py's loop inside a 1-rank RCCL group, collective forced
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    ex = LoglikeExchange(D, dev, force_collective=True)
    assert ex._force and ex.world == 1
    got = []

    def consume(full):
        if full is not None:
            got.append(full.clone())

    run, drain = bench.make_runner(graph, si, ex, consume)
    for i in range(steps):
        set_leaves(i)
        run(i)
    drain()
    torch.cuda.synchronize()
    assert len(got) == steps, len(got)
    for i in range(steps):
        assert torch.equal(got[i], want[i]), (i, float((got[i] - want[i]).abs().max()))
    assert not torch.equal(want[0], want[steps - 1])
    # the synchronous form, straight from the static output
    set_leaves(3)
    out = graph()
    full = ex(out[si])
    torch.cuda.synchronize()
    assert torch.equal(full, want[3])
finally:
    dist.destroy_process_group()
print("NCCL_ONE_RANK_OK", cfg, D, steps)
"""


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("cfg,D,steps", [("c2", 1024, 60), ("c5", 128, 50)])
def test_real_steps_through_one_rank_rccl_group(cfg, D, steps):
    code = WORKER.format(root=ROOT, cfg=cfg, D=D, steps=steps, port=_free_port())
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "NCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_under_the_launcher_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` (the driver's form for N > 1, with one
    rank) with the collective forced: the JSON line comes out, the exchange ran inside the timed region"""
    import bench

    argv = bench.launcher_argv(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "c4", "--global-draws", "64",
                                   "--steps", "20", "--warmup", "3", "--no-extras", "--no-cpu-baseline", "--no-stats"])
    env = _env()
    env["EXO_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run(argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["scaling"] == "strong"
    assert line["config"]["global_draws"] == 64 and line["value"] > 0
