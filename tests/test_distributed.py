"""CPU, world_size 2, gloo: the draw-sharding path (shard -> local evaluation ->
one all-reduce of the per-draw log-likelihood vector).  The local evaluator here
is the ORACLE (checker role only: there is no GPU in this job); on the GPU box the
same module is driven by the HIP ops (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_draw, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from exoplanet_amd.distributed import shard_bounds, sharded_log_likelihood
        from oracle import numpy_port as P

        rng = np.random.default_rng(0)
        t = np.arange(600) * (2.0 / 1440.0) + 0.9
        y = 5e-4 * rng.normal(size=t.size)
        r = torch.tensor(0.1 * (1 + 0.05 * rng.normal(size=n_draw)), dtype=torch.float64)

        def loglike(params):
            # white-noise log-likelihood of each local draw, evaluated by the oracle
            out = []
            for rv in params["r"].numpy():
                f = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(
                    orbit=P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1), r=rv, t=t)[:, 0]
                out.append(-0.5 * np.sum(((y - f) / 5e-4) ** 2))
            return torch.tensor(out, dtype=torch.float64)

        full, local = sharded_log_likelihood(loglike, {"r": r}, n_draw)
        lo, hi = shard_bounds(n_draw)
        assert local.shape[0] == hi - lo
        assert torch.equal(full[lo:hi], local)
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
        if rank == 0:
            ref = loglike({"r": r}).numpy()
            np.save(os.path.join(out_dir, "ref.npy"), ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_draw", [8, 7])
def test_draw_sharding_world2(tmp_path, n_draw):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_draw, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "full_0.npy")
    b = np.load(tmp_path / "full_1.npy")
    ref = np.load(tmp_path / "ref.npy")
    assert a.shape == (n_draw,)
    np.testing.assert_array_equal(a, b)          # every rank sees the same full vector
    np.testing.assert_allclose(a, ref, rtol=1e-13)


def _worker_ttv(rank, world, port, n_draw, out_dir):
    """draws that differ in their transit-timing offsets: the per-draw tables are one more entry
    of the sharded parameter tree (a list of (n_draw, n_transit) tensors, one per planet)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from exoplanet_amd.distributed import shard_bounds, sharded_log_likelihood
        from oracle import numpy_port as P

        rng = np.random.default_rng(1)
        t = np.linspace(0.0, 30.0, 900)
        y = 5e-4 * rng.normal(size=t.size)
        params = {"r": torch.tensor(0.1 * (1 + 0.05 * rng.normal(size=n_draw)), dtype=torch.float64),
                  "ttvs": [torch.tensor(0.02 * rng.normal(size=(n_draw, 8)), dtype=torch.float64)]}

        def loglike(local):
            out = []
            for rv, ttv in zip(local["r"].numpy(), local["ttvs"][0].numpy()):
                orbit = P.TTVOrbit(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3]), ttvs=[ttv])
                f = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=np.array([rv]), t=t)[:, 0]
                out.append(-0.5 * np.sum(((y - f) / 5e-4) ** 2))
            return torch.tensor(out, dtype=torch.float64)

        full, local = sharded_log_likelihood(loglike, params, n_draw)
        lo, hi = shard_bounds(n_draw)
        assert local.shape[0] == hi - lo and torch.equal(full[lo:hi], local)
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "ref.npy"), loglike(params).numpy())
    finally:
        dist.destroy_process_group()


def test_draw_sharding_with_timing_tables_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker_ttv, args=(2, port, 5, str(tmp_path)), nprocs=2, join=True)
    a, b, ref = (np.load(tmp_path / f) for f in ("full_0.npy", "full_1.npy", "ref.npy"))
    np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(a, ref, rtol=1e-13)
    assert np.ptp(ref) > 0          # the draws really differ


def _worker_bench_exchange(rank, world, port, n_global, out_dir):
    """bench.py's own multi-GPU step function (the collective part: exchange_step on a
    LoglikeExchange), equal shards -> all_gather_into_tensor, ragged -> all-reduce; and the same
    over a sub-group whose ranks differ from the default group's (shard bounds must
    come from the group the collective runs on)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from exoplanet_amd.distributed import LoglikeExchange, gather_loglike, shard_bounds

        lo, hi = shard_bounds(n_global)
        ex = LoglikeExchange(n_global, torch.device("cpu"))
        assert ex.equal == (n_global % world == 0)
        for it in range(3):      # the buffer is reused step after step
            local = torch.arange(lo, hi, dtype=torch.float64) * (it + 1) + 0.25
            full = bench.exchange_step(ex, local)
            want = torch.arange(n_global, dtype=torch.float64) * (it + 1) + 0.25
            assert torch.equal(full, want), (rank, it, full, want)
        with pytest.raises(ValueError):
            ex(torch.zeros(hi - lo + 1, dtype=torch.float64))
        # the pipelined form: the source may be overwritten right after start(); finish() returns the latest step's vector
        assert ex.finish() is None
        src = torch.zeros(hi - lo, dtype=torch.float64)
        consume = bench.ExchangeConsumer(n_global, torch.device("cpu"))
        for it in range(5):
            src.copy_(torch.arange(lo, hi, dtype=torch.float64) * (it + 2) - 0.5)
            prev = bench.exchange_step(ex, src, pipelined=True)
            src.fill_(float("nan"))                      # (what the next graph replay does to its static output)
            # every start() hands back the PREVIOUS step's completed vector: a consumer sees each step once
            if it == 0:
                assert prev is None
            else:
                assert torch.equal(prev, torch.arange(n_global, dtype=torch.float64) * (it + 1) - 0.5), (rank, it, prev)
            consume(prev)
        last = ex.finish()
        assert torch.equal(last, torch.arange(n_global, dtype=torch.float64) * 6 - 0.5)
        consume(last)
        assert consume.steps == 5
        want_sum = sum(torch.arange(n_global, dtype=torch.float64) * (it + 2) - 0.5 for it in range(5))
        assert torch.allclose(consume.acc, want_sum, rtol=1e-14)
        with pytest.raises(ValueError):
            ex.start(torch.zeros(hi - lo + 1, dtype=torch.float64))
        # a group with the ranks in reverse order: position in the group != global rank
        grp = dist.new_group(ranks=list(range(world))[::-1])
        glo, ghi = shard_bounds(n_global, group=grp)
        assert (glo, ghi) == shard_bounds(n_global, rank=dist.get_rank(grp), world=world)
        out = gather_loglike(torch.arange(glo, ghi, dtype=torch.float64), n_global, group=grp)
        assert torch.equal(out, torch.arange(n_global, dtype=torch.float64))
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [8, 7])
def test_bench_exchange_step_world2(tmp_path, n_global):
    port = _free_port()
    mp.spawn(_worker_bench_exchange, args=(2, port, n_global, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0.npy").exists() and (tmp_path / "ok_1.npy").exists()


def test_exchange_single_process():
    from exoplanet_amd.distributed import LoglikeExchange, shard_bounds

    assert shard_bounds(10, rank=1, world=3) == (4, 7)
    ex = LoglikeExchange(5, torch.device("cpu"))
    x = torch.arange(5, dtype=torch.float64)
    assert torch.equal(ex(x), x) and ex(x) is ex.out


def _worker_runner(rank, world, port, n_global, out_dir):
    """bench.py's timed loop (make_runner + time_steps) for a strong-scaling config under gloo: a stand-in step whose
    per-draw scalar is known in closed form, ragged and equal shards; every step's full vector reaches the consumer
    exactly once, in order, on every rank"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from exoplanet_amd.distributed import LoglikeExchange, shard_bounds

        lo, hi = shard_bounds(n_global, rank, world)       # what main() does with --global-draws
        ex = LoglikeExchange(n_global, torch.device("cpu"))
        consume = bench.ExchangeConsumer(n_global, torch.device("cpu"))
        state = {"k": 0}
        static = torch.zeros(hi - lo, dtype=torch.float64)     # (a replayed graph's static output: overwritten every step)

        def step_fn():
            state["k"] += 1
            static.copy_(torch.arange(lo, hi, dtype=torch.float64) + 100.0 * state["k"])
            return (None, static)

        run, drain = bench.make_runner(step_fn, 1, ex, consume)
        K = 7
        for i in range(K):
            run(i)
        drain()
        assert consume.steps == K
        want = K * torch.arange(n_global, dtype=torch.float64) + 100.0 * sum(range(1, K + 1))
        assert torch.equal(consume.acc, want), (rank, consume.acc, want)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [64, 7])
def test_bench_runner_world2(tmp_path, n_global):
    port = _free_port()
    mp.spawn(_worker_runner, args=(2, port, n_global, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0.npy").exists() and (tmp_path / "ok_1.npy").exists()


def test_bench_config_table():
    """--config: every BASELINE GPU config has a builder; C4 / C5 default to their 8-GPU totals"""
    import bench

    assert sorted(bench.WORKLOADS) == ["c2", "c3", "c4", "c5"]
    assert bench.DEFAULT_GLOBAL_DRAWS == {"c4": 512, "c5": 1024}
    from exoplanet_amd.distributed import shard_bounds
    assert [shard_bounds(512, r, 8)[1] - shard_bounds(512, r, 8)[0] for r in range(8)] == [64] * 8
    assert [shard_bounds(1024, r, 8)[1] - shard_bounds(1024, r, 8)[0] for r in range(8)] == [128] * 8


def test_bench_turns_itself_into_the_launcher(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE (the form of the driver's recorded command) re-executes itself under
    torch.distributed.run on the loopback address with the same arguments (it used to exit)"""
    import sys

    import bench

    argv = bench.launcher_argv(4, ["bench.py", "--gpus", "4", "--steps", "7", "--config", "c5"])
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(argv[argv.index("--master-port") + 1]) < 65536
    k = argv.index(os.path.abspath("bench.py"))
    assert argv[k + 1:] == ["--gpus", "4", "--steps", "7", "--config", "c5"]
    # main() hands over before it touches a GPU
    seen = {}

    def fake_execv(path, args):
        seen["path"], seen["args"] = path, args
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("EXO_BENCH_NO_REEXEC", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["path"] == sys.executable and "--nproc-per-node=2" in seen["args"]
    # a launcher that set WORLD_SIZE=1 for --gpus 2 is a mistake, not a reason to spawn again
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value)
