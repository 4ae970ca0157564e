"""GPU: bench.py with TWO ranks on the box's one GPU (EXO_BENCH_SHARE_GPU=1: devices by rank % count, the per-step collective
over gloo -- RCCL wants a device per rank): the whole N-rank control flow of the contract -- shards of the draws, the pipelined
exchange of the per-draw scalars, barrier + synchronize on both sides of the timed steps, the max over ranks, ONE line from
rank 0 -- launched the way the driver launches it.  The aggregate is one GPU's, shared: the line's bookkeeping is what is held."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, EXO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras", "--no-stats"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_c2():
    out = _run(["--draws-per-gpu", "128"], 29641)
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["scaling"] == "weak"
    assert out["config"]["global_draws"] == 256 and out["config"]["draws_per_gpu"] == 128
    assert out["value"] > 0 and abs(out["value"] - 256 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert out["roofline"]["bound"] == "hbm"
    assert len(json.dumps(out)) < 4096


def test_two_ranks_strong_scaling_c5():
    out = _run(["--config", "c5", "--global-draws", "64"], 29642)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_draws"] == 64
    assert out["value"] > 0
