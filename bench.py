#!/usr/bin/env python
"""bench.py -- light-curve evaluations / s (value + gradient) at 150 000 cadences.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run, one rank per GPU (RCCL).  Rank 0
prints ONE JSON line.

Workload = BASELINE.json configs[1] ("C2"): single planet, e = 0.3,
omega = 1.1, P = 3.5 d, t0 = 1, b = 0.3, r = 0.1, (u1, u2) = (0.3, 0.2), 150 000
two-minute cadences, EVERY cadence evaluated (use_in_transit=False, the
roofline case), float64, cotangent gbar ~ N(0,1).  One *evaluation* = forward
flux for all 150 000 cadences of one posterior draw + the VJP of gbar back to
all orbit / limb-darkening parameters.  One *step* = one pass of the hot path
over a batch of `--draws-per-gpu` draws (base parameters x (1 + 1e-3 N(0,1))):
leaf parameters -> record-packing kernel (KeplerianOrbit algebra + get_cl) ->
scan + heavy kernels (value + VJP in one sweep) -> packing VJP -> leaf gradients; with N > 1 ranks each own
their draws (weak scaling) and exchange only the per-draw scalar sum(gbar*flux)
by one all-reduce.  Inputs are resident in HBM before the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAD = 150_000
CADENCE = 2.0 / 1440.0
ALG_BYTES_PER_UNIT = 24          # read t 8 + read gbar 8 + write flux 8 per (draw, cadence), SURVEY.md 8d
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def hip_runtime():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64 not found")


class HipEvents:
    """K (start, stop) hipEvent pairs recorded by the C ABI around the dominant kernel."""

    def __init__(self, k):
        self.hip = hip_runtime()
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.pairs = []
        for _ in range(k):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0
            assert self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def handles(self, i):
        a, b = self.pairs[i]
        return a.value, b.value

    def mean_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            out.append(ms.value)
        return float(np.mean(out)), out


def measured_traffic(n_draw):
    """HBM bytes per sweep (scan + heavy kernels) from the committed PMC passes
    (profiles/r01_pmc.json), scaled to this run's draw count; None if absent."""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
        tot = sum(2.0 * k["fetch_kib"] + k["write_kib"] for k in p["kernels"].values())
        return tot * 1024.0 * n_draw / p["draws"]
    except Exception:
        return None


def make_leaves(n_draw, seed, dev):
    """C2 base parameters x (1 + 1e-3 N(0,1)), one row per draw, as autograd leaves."""
    rng = np.random.default_rng(seed)
    base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1, u1=0.3, u2=0.2)
    leaves = {}
    for k, v in base.items():
        x = v * (1 + 1e-3 * rng.normal(size=(n_draw, 1)))
        if k == "ecc":
            x = np.clip(x, 0.0, 0.95)
        if k in ("u1", "u2"):
            x = x[:, 0]
        leaves[k] = torch.tensor(x, dtype=torch.float64, device=dev, requires_grad=True)
    return leaves


def step(xo, ops, leaves, t, gbar, events=(None, None), use_in_transit=False):
    """one pass of the hot path over the batch: returns (flux, L[d], grads of the leaves)"""
    orbit = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                              omega=leaves["omega"])
    rec, ld, _, flags = orbit.kernel_inputs(leaves["r"], (leaves["u1"], leaves["u2"]), use_in_transit=use_in_transit)
    flux, L = ops.transit_flux_dot(t, rec, ld, gbar, flags=flags, events=events)
    grads = torch.autograd.grad(L.sum(), list(leaves.values()))
    return flux, L, grads


def time_steps(fn, steps, warmup, dist, dev):
    for _ in range(warmup):
        fn(-1)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def cpu_baseline(seconds_target=12.0):
    """The oracle's C port (scalar, 1 core) on a bounded sample of the same workload."""
    from oracle import c_port as C
    from oracle import numpy_port as P

    rng = np.random.default_rng(2)
    t = np.arange(N_CAD) * CADENCE
    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec = np.zeros((1, 1, P.NPAR))
    rec[0, 0, [P.P_N, P.P_TP, P.P_ECC, P.P_COSW, P.P_SINW, P.P_COSI, P.P_SINI, P.P_AOR, P.P_ROR]] = [
        orbit.n[0], orbit.t_periastron[0], 0.3, np.cos(1.1), np.sin(1.1), orbit.cos_incl[0], orbit.sin_incl[0],
        orbit.a[0], 0.1]
    rec[0, 0, [P.P_T0, P.P_PERIOD, P.P_TS, P.P_TE, P.P_TS2, P.P_TE2]] = [1.0, 3.5, -np.inf, np.inf, -np.inf, np.inf]
    c = P.get_cl(0.3, 0.2)[None]
    g = rng.normal(size=(1, N_CAD))
    C.transit(t, rec, c, g)  # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        C.transit(t, rec, c, g)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{n} evaluations (value+VJP, every cadence) of the {N_CAD}-cadence C2 system, "
                      f"oracle/c scalar port, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--draws-per-gpu", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("EXO_BENCH_FORCE_DIST") == "1":   # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod

    import exoplanet_amd as xo
    from exoplanet_amd import ops, _lib

    _lib.load()  # fail loudly if the HIP library is missing
    D = args.draws_per_gpu
    t = torch.arange(N_CAD, dtype=torch.float64, device=dev) * CADENCE
    gbar = torch.as_tensor(np.random.default_rng(2 + rank).normal(size=(D, N_CAD)), device=dev)
    leaves = make_leaves(D, 100 + rank, dev)
    L_all = torch.zeros(world * D, dtype=torch.float64, device=dev)
    events = HipEvents(args.steps)

    def one(i, use_in_transit=False, ev=True):
        evs = events.handles(i) if (ev and i >= 0) else (None, None)
        flux, L, grads = step(xo, ops, leaves, t, gbar, events=evs, use_in_transit=use_in_transit)
        if dist is not None:
            L_all.zero_()
            L_all[rank * D:(rank + 1) * D] = L.detach()
            dist.all_reduce(L_all)     # the only collective: per-draw scalars, 8 B x D x N
        return flux, L, grads

    # The step is a dozen short launches (packing kernel, scan, heavy, reduce, packing
    # VJP and a few tensor-shuffling torch kernels): launch-bound when issued eagerly,
    # so the timed region replays it as ONE hipGraph (the collective stays outside).
    graph = None
    static = {}
    if not args.no_graph:
        names = list(leaves)
        graph = xo.GraphedStep(lambda *vals: step(xo, ops, dict(zip(names, vals)), t, gbar), *leaves.values())
        static["flux"], static["L"], static["grads"] = graph.outputs

    def one_graph(i):
        graph()
        if dist is not None:
            L_all.zero_()
            L_all[rank * D:(rank + 1) * D] = static["L"]
            dist.all_reduce(L_all)

    wall = time_steps(one_graph if graph is not None else one, args.steps, args.warmup, dist, dev)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    if graph is not None:
        # hipEvents cannot bracket a node inside a replayed graph: time the dominant
        # kernel over the same number of directly launched steps, same inputs
        time_steps(lambda i: one(i), args.steps, 1, None, dev)
    kernel_ms, _ = events.mean_ms()

    out = None
    if rank == 0:
        evals = world * D * args.steps
        alg_bytes = ALG_BYTES_PER_UNIT * D * N_CAD
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "light-curve evals/sec (value+grad) at 150k cadences",
            "value": evals / wall,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] (C2): single planet e=0.3 Kepler solve + quadratic limb-darkened "
                            "transit, 150000 cadences, value+grad, every cadence evaluated (use_in_transit=False)",
                "n_cadences": N_CAD, "draws_per_gpu": D, "global_draws": world * D,
                "parallelism": f"draws sharded over {world} GPU(s); all-reduce of per-draw scalars only",
                "step": "leaf params -> record-packing kernel (orbit algebra + get_cl) -> scan + heavy kernels "
                        "(value+VJP, one sweep) -> packing VJP kernel -> leaf gradients" + ("; replayed as one hipGraph" if graph is not None else "; eager launches"),
            },
            "roofline": {
                "bound": "hbm", "kernel": "transit_window_kernel + transit_scan_kernel + transit_heavy_kernel + transit_vjp_reduce_kernel (one sweep)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(D),
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms,
                "traffic_frac": (measured_traffic(D) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if measured_traffic(D) else None,
                "note": "achieved = 24 B per (draw, cadence) x draws x cadences / mean hipEvent time of the launches "
                        "that make up the one algorithmic sweep (window constants, scan = classify + zero-fill, "
                        "heavy = active cadences, reduce = block partials).  The algorithmic count charges a gbar "
                        "read to every cadence; the kernels read it only for the ~3 % that are in transit, so "
                        "achieved can exceed peak.  traffic = PMC-measured HBM bytes per sweep (profiles/), "
                        "traffic_frac = traffic / time / peak: the store stream of zeros is what bounds the scan "
                        "kernel, VALU latency bounds the heavy one.  rocprof per-kernel averages: profiles/",
            },
        }

    # the extra legs are single-GPU diagnostics: under torch.distributed.run they would only add
    # barriers that every rank has to reach (an exception on one rank would hang the others)
    if not args.no_extras and world == 1:
        # reference-default semantics (use_in_transit=True): same step, windows on.  Not `value`.
        ex_steps = max(5, args.steps // 2)
        ev2 = HipEvents(ex_steps)
        events_backup, events = events, ev2

        def one_win(i):
            evs = ev2.handles(i) if i >= 0 else (None, None)
            return step(xo, ops, leaves, t, gbar, events=evs, use_in_transit=True)

        wall2 = time_steps(one_win, ex_steps, 2, dist, dev)
        k2, _ = ev2.mean_ms()
        # op-level time of the one-sweep kernel call alone (no torch glue)
        orbit = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                                  omega=leaves["omega"])
        rec, c, _, _ = orbit.kernel_inputs(leaves["r"], (leaves["u1"], leaves["u2"]))
        rec, c = rec.detach(), c.detach()
        wall3 = time_steps(lambda i: ops.transit_flux_value_and_vjp(t, rec, c, gbar), ex_steps, 2, dist, dev)
        # BASELINE configs[2] (C3): the same light curve + a celerite SHO-term GP log-likelihood on
        # the residual, value + gradient w.r.t. the orbit / limb-darkening leaves (user-level API)
        c3 = None
        try:
            yobs = 5e-4 * torch.randn(N_CAD, dtype=torch.float64, device=dev)
            ones = torch.ones(D, dtype=torch.float64, device=dev)
            names = list(leaves)

            def one_c3(i):
                orbit3 = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                                           omega=leaves["omega"])
                lc = xo.LimbDarkLightCurve(leaves["u1"], leaves["u2"]).get_light_curve(orbit=orbit3, r=leaves["r"], t=t)
                gp = xo.gp.GaussianProcess(xo.gp.terms.SHOTerm(sigma=1e-3 * ones, rho=5.0 * ones, Q=0.7071 * ones),
                                           t=t, yerr=5e-4, mean=lc.sum(-1))
                ll = gp.log_likelihood(yobs)
                return torch.autograd.grad(ll.sum(), [leaves[k] for k in names])

            c3_steps = 4
            wall_c3 = time_steps(one_c3, c3_steps, 2, dist, dev)
            c3 = {"evals_per_s": world * D * c3_steps / wall_c3, "ms_per_step": 1e3 * wall_c3 / c3_steps,
                  "note": "C3: C2 light curve + SHO-term celerite GP on the residual (recurrences in parallel "
                          "over time), value + gradient, eager launches"}
        except Exception as exc:  # an extra leg must not take the headline measurement down with it
            c3 = {"error": repr(exc)[:200]}
        # the same C2 step with transit-timing variations -- 60 labelled transits, every
        # draw its own offsets (gradient to each of them) -- user-level TTVOrbit, replayed as a hipGraph
        ttv = None
        try:
            n_tr = int((float(t[-1]) - 1.0) / 3.5) + 1
            offs = torch.tensor(0.01 * np.random.default_rng(7).normal(size=(D, n_tr)), dtype=torch.float64,
                                device=dev, requires_grad=True)
            tnames = list(leaves) + ["ttvs"]

            def ttv_step(*vals):
                Lv = dict(zip(tnames, vals))
                orb = xo.orbits.TTVOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"],
                                         ttvs=[Lv["ttvs"]])
                rec_t, ld_t, _, fl = orb.kernel_inputs(Lv["r"], (Lv["u1"], Lv["u2"]))
                ed, sh = orb.kernel_ttv()
                _, dot = ops.transit_flux_dot(t, rec_t, ld_t, gbar, flags=fl, ttv=(ed.contiguous(), sh.contiguous()))
                return (dot.detach(),) + torch.autograd.grad(dot.sum(), vals)

            tg = xo.GraphedStep(ttv_step, *leaves.values(), offs)
            wall_ttv = time_steps(lambda i: tg(), ex_steps, 2, dist, dev)
            ttv = {"evals_per_s": world * D * ex_steps / wall_ttv, "ms_per_step": 1e3 * wall_ttv / ex_steps,
                   "transits": n_tr,
                   "note": "C2 step with a TTVOrbit (timing tables in the fused kernels, gradients to every "
                           "per-transit offset), hipGraph replay"}
        except Exception as exc:
            ttv = {"error": repr(exc)[:200]}
        if rank == 0:
            out["extras"] = {
                "c3_light_curve_plus_sho_gp": c3,
                "c2_with_transit_timing_variations": ttv,
                "in_transit_only": {"evals_per_s": world * D * ex_steps / wall2, "kernel_ms": k2,
                                    "alg_GBps": ALG_BYTES_PER_UNIT * D * N_CAD / (k2 * 1e-3) / 1e9,
                                    "note": "reference default use_in_transit=True (contact-point windows); this "
                                            "extra leg launches eagerly (no hipGraph), so its evals/s is launch-bound"},
                "op_level_every_cadence": {"evals_per_s": world * D * ex_steps / wall3,
                                           "ms_per_step": 1e3 * wall3 / ex_steps,
                                           "note": "fused kernel call only, no orbit algebra / autograd"},
            }
        events = events_backup

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (buffered when piped): push it out first so
        # that the JSON line is the last line of stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
