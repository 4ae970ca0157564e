#!/usr/bin/env python
"""bench.py -- light-curve evaluations / s (value + gradient) at 150 000 cadences.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run, one rank per GPU (RCCL).  Rank 0
prints ONE JSON line.

Workload = BASELINE.json configs[1] ("C2"): single planet, e = 0.3,
omega = 1.1, P = 3.5 d, t0 = 1, b = 0.3, r = 0.1, (u1, u2) = (0.3, 0.2), 150 000
two-minute cadences, float64, cotangent gbar ~ N(0,1), use_in_transit=False: the
output is the DENSE flux array [draws][150 000]; every cadence is classified on
the device (binary search of the conjunction windows in the sorted time array),
the ~3 % that can overlap the stellar disk are solved (Kepler + solution vector
+ reverse sweep), the rest are written as zeros.  One
*evaluation* = forward flux for all 150 000 cadences of one posterior draw + the
VJP of gbar back to all orbit / limb-darkening parameters.  One *step* = one
pass of the hot path over a batch of `--draws-per-gpu` draws (base parameters x
(1 + 1e-3 N(0,1))): leaf parameters -> record-packing kernel (KeplerianOrbit
algebra + get_cl) -> window + run-enumeration + heavy + finish kernels (value +
VJP in one sweep; the heavy kernel zero-fills the dense flux array while it
solves) -> packing VJP -> leaf gradients, replayed as one hipGraph.  With N > 1
ranks each own their draws and exchange only the per-draw scalar sum(gbar*flux):
ONE collective per step (exoplanet_amd.distributed.LoglikeExchange).  Default is
weak scaling (fixed draws per GPU); `--global-draws G` fixes the total instead
(BASELINE C4 / C5: 512 / 1024 draws over 8 GPUs) and reports "strong".
Inputs are resident in HBM before the timed region.

Timing: `value` comes from EXACTLY --steps steps between barrier + synchronize
on both sides (max over ranks).  Independently of --steps, `timing` reports
median / p10 / p90 per step over >= 100 steps and >= 2 s.
"""
import argparse
import ctypes
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAD = 150_000
CADENCE = 2.0 / 1440.0
SURVEY_BYTES_PER_UNIT = 24       # SURVEY.md 8d count: read t 8 + read gbar 8 + write flux 8 per (draw, cadence)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def hip_runtime():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64 not found")


class HipEvents:
    """K (start, stop) hipEvent pairs recorded by the C ABI around the kernels of one sweep,
    on the stream they are launched on."""

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

    def times_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            out.append(ms.value)
        return out

    def mean_ms(self):
        v = self.times_ms()
        return float(np.mean(v)), v


def quantiles(ms):
    ms = np.asarray(ms, dtype=np.float64)
    return {"iters": int(ms.size), "median_ms": float(np.median(ms)), "p10_ms": float(np.percentile(ms, 10)),
            "p90_ms": float(np.percentile(ms, 90)), "mean_ms": float(ms.mean())}


def stats_loop(fn, dev, n_iters, chunk=100):
    """per-step durations of n_iters back-to-back steps (torch events on the current stream: the
    steps are launched on it).  The count is fixed up front -- under torch.distributed every rank
    must issue the same number of collectives."""
    ms = []
    stream = torch.cuda.current_stream(dev)
    while len(ms) < n_iters:
        k = min(chunk, n_iters - len(ms))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        evs[0].record(stream)
        for i in range(k):
            fn(-1)
            evs[i + 1].record(stream)
        torch.cuda.synchronize(dev)
        ms.extend(evs[i].elapsed_time(evs[i + 1]) for i in range(k))
    return quantiles(ms)


def measured_traffic(n_draw):
    """HBM bytes per sweep from this round's committed PMC passes (profiles/r02_pmc.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per the gfx950 note), scaled to
    this run's draw count.  A cross-reference, not a live measurement: None if the file is absent
    or was taken on another kernel generation."""
    try:
        import hashlib

        p = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc.json")))
        src = os.path.join(ROOT, "exoplanet_amd", "csrc", "exo_transit.hip")
        if p.get("kernel_source_sha256") != hashlib.sha256(open(src, "rb").read()).hexdigest():
            return None                      # the counters were taken on other kernels: say nothing rather than something stale
        tot = sum(2.0 * k["fetch_kib"] + k["write_kib"] for k in p["kernels"].values() if k.get("dispatches_fetch", 0) > 5)
        return {"bytes": tot * 1024.0 * n_draw / p["draws"], "source": "profiles/r02_pmc.json (same exo_transit.hip)",
                "draws_profiled": p["draws"]}
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


_ONES = {}


def step(xo, ops, leaves, t, gbar, events=(None, None), use_in_transit=False, **kw):
    """one pass of the hot path over the batch: returns (flux, L[d], grads of the leaves).  The leaves go to the
    kernels as they are (KeplerianOrbit.flux_dot: column-form packing kernel -> sweep -> packing VJP with the
    cotangent of L folded in); the cotangent of L is a constant vector of ones (d sum(L) / d leaves)."""
    orbit = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                              omega=leaves["omega"])
    flux, L = orbit.flux_dot(leaves["r"], (leaves["u1"], leaves["u2"]), t, gbar, use_in_transit=use_in_transit,
                             events=events, **kw)
    key = (L.shape[0], L.device)
    if key not in _ONES:
        _ONES[key] = torch.ones(L.shape[0], dtype=L.dtype, device=L.device)
    grads = torch.autograd.grad(L, list(leaves.values()), grad_outputs=_ONES[key])
    return flux, L, grads


def exchange_step(exchange, L_local, pipelined=False):
    """the multi-GPU part of a step: one collective, every rank ends up with all per-draw scalars.
    ``pipelined``: the collective is issued asynchronously from a private copy of the rank's scalars and overlaps the
    next step's kernels (LoglikeExchange.start); the caller ends the loop with ``exchange.finish()``.
    (tests/test_distributed.py drives this function on CPU under gloo, world size 2.)"""
    if pipelined and not _SYNC_EXCHANGE[0]:
        try:
            return exchange.start(L_local)
        except RuntimeError as e:      # (a backend without asynchronous collectives: say so once, go on synchronously)
            print(f"bench.py: asynchronous exchange failed ({e}); using the synchronous collective", file=sys.stderr)
            _SYNC_EXCHANGE[0] = True
    return exchange(L_local)


_SYNC_EXCHANGE = [os.environ.get("EXO_BENCH_SYNC_EXCHANGE") == "1"]


def time_steps(fn, steps, warmup, dist, dev, drain=None):
    """``drain``: called after the last step, inside the timed region (collectives still in flight are waited for)"""
    for _ in range(warmup):
        fn(-1)
    if drain is not None:
        drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    if drain is not None:
        drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


# ------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle's C port -- the thing being CHECKED AGAINST
# elsewhere, timed here beside the GPU number; never part of `value`.
# ------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def usable_cpus():
    """(threads this process can actually run at once, how that was decided): the affinity mask, capped by the
    cgroup CPU quota -- the GPU boxes show 256 logical cores but grant 16 cores' worth of time (cpu.max), and 256
    OpenMP threads on that only measure the throttling"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    why = f"affinity mask: {n} of {os.cpu_count()} logical cores"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:                                                                  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota))
        why = f"cgroup CPU quota: {quota:g} cores' worth of time on {os.cpu_count()} logical cores"
    return n, why


def cpu_baseline(budget_s=24.0):
    """oracle/c on the host cores of this box, on a bounded sample of the C2 workload:
    (i) 1 core, every cadence evaluated (what the reference does with use_in_transit=False);
    (ii) 1 core, in-transit cadences only (the reference's default use_in_transit=True);
    (iii) all cores, one draw per thread (PyMC's one process per chain on every core), every cadence;
    (iv) all cores, in-transit only."""
    from oracle import c_port as C
    from oracle import numpy_port as P

    # the library travels prebuilt; rebuild it for THIS host's cores (-march=native) when a compiler is here
    try:
        subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle", "c")], check=True, timeout=120,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        pass
    lib = C.lib()
    usable, usable_why = usable_cpus()
    n_threads = int(os.environ.get("EXO_BENCH_CPU_THREADS", usable))
    rng = np.random.default_rng(2)
    t = np.arange(N_CAD) * CADENCE
    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)

    def records(n_draw, window):
        ts, te = (-np.inf, np.inf)
        rec = np.zeros((n_draw, 1, P.NPAR))
        rec[:, 0, [P.P_N, P.P_TP, P.P_ECC, P.P_COSW, P.P_SINW, P.P_COSI, P.P_SINI, P.P_AOR, P.P_ROR]] = [
            orbit.n[0], orbit.t_periastron[0], 0.3, np.cos(1.1), np.sin(1.1), orbit.cos_incl[0], orbit.sin_incl[0],
            orbit.a[0], 0.1]
        if window:   # first / fourth contact relative to t0 (keplerian.py:744-763)
            Ml, Mr, flag = P.contact_points(orbit.a, orbit.ecc, orbit.cos_omega, orbit.sin_omega, orbit.cos_incl,
                                            orbit.sin_incl, orbit.r_star + 0.1)
            assert np.all(flag == 0)
            hp = 0.5 * orbit.period
            ts = float(np.ravel(np.mod((Ml - orbit.M0) / orbit.n + hp, orbit.period) - hp)[0])
            te = float(np.ravel(np.mod((Mr - orbit.M0) / orbit.n + hp, orbit.period) - hp)[0])
            ts = ts - 3.5 if ts > 0 else ts
            te = te + 3.5 if te < 0 else te
        rec[:, 0, [P.P_T0, P.P_PERIOD, P.P_TS, P.P_TE, P.P_TS2, P.P_TE2]] = [1.0, 3.5, ts, te, -np.inf, np.inf]
        rec[:, 0, P.P_ROR] *= 1 + 1e-3 * rng.normal(size=n_draw)
        return rec

    def leg(n_draw, threads, window, seconds):
        lib.oracle_set_threads(int(threads))
        rec = records(n_draw, window)
        c = np.repeat(P.get_cl(0.3, 0.2)[None], n_draw, 0)
        g = rng.normal(size=(n_draw, N_CAD))
        C.transit(t, rec, c, g, window=window)  # warm
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            C.transit(t, rec, c, g, window=window)
            n += n_draw
        dt = time.perf_counter() - t0
        return {"evals_per_s": n / dt, "evals": n, "seconds": dt, "threads": int(threads),
                "semantics": "in-transit cadences only (use_in_transit=True)" if window else
                             "every cadence solved (use_in_transit=False)"}

    share = budget_s / 4.0
    legs = {"one_core_every_cadence": leg(1, 1, False, share)}
    try:
        legs["one_core_in_transit"] = leg(1, 1, True, share)
    except Exception as exc:      # window helper missing in the oracle: report, do not die
        legs["one_core_in_transit"] = {"error": repr(exc)[:160]}
    legs["all_cores_every_cadence"] = leg(n_threads, n_threads, False, share)
    try:
        legs["all_cores_in_transit"] = leg(4 * n_threads, n_threads, True, share)
    except Exception as exc:
        legs["all_cores_in_transit"] = {"error": repr(exc)[:160]}
    lib.oracle_set_threads(1)
    one = legs["one_core_every_cadence"]
    return {"value": one["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{one['evals']} evaluations (value+VJP, every cadence solved) of the {N_CAD}-cadence C2 "
                      f"system, oracle/c scalar port, {one['seconds']:.1f} s on 1 of {os.cpu_count()} host cores; "
                      f"legs: the same on all {n_threads} usable cores ({usable_why}; OpenMP, one draw per thread) and "
                      "with the reference's default in-transit selection",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "usable_cores": usable, "usable_cores_from": usable_why,
            "legs": legs,
            "note": "the reference's own Ops (exoplanet_core, celerite2) are not installable here: kind = port"}


# ------------------------------------------------------------------------------------------
# extras (single GPU diagnostics; never `value`)
# ------------------------------------------------------------------------------------------
def time_events(fn, dev, iters, warmup=2):
    """mean / quantiles of `fn` per call, torch events on the current stream"""
    for _ in range(warmup):
        fn()
    stream = torch.cuda.current_stream(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record(stream)
    for i in range(iters):
        fn()
        evs[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    return quantiles([evs[i].elapsed_time(evs[i + 1]) for i in range(iters)])


def extra_ops(ops, dev, n=150_000_000):
    """the reference's standalone Ops at n = 1.5e8 elements: GB/s against
    their algorithmic bytes -- kepler 2 in + 2 out = 32 B/elt; quad_solution_vector 2 in + 3 out =
    40 B/elt (value) or 2 in + 9 out = 88 B/elt (with ds/db, ds/dr)"""
    out = {}
    g = torch.Generator(device=dev).manual_seed(11)
    M = (torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5) * 800.0
    e = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 0.9
    with torch.no_grad():
        q = time_events(lambda: ops.kepler(M, e), dev, 10)
    out["kepler"] = {"n": n, **q, "bytes_per_elt": 32, "GBps": 32.0 * n / (q["median_ms"] * 1e-3) / 1e9,
                     "frac_of_hbm_peak": 32.0 * n / (q["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del M, e
    b = torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 1.3
    r = torch.full((n,), 0.1, dtype=torch.float64, device=dev)
    with torch.no_grad():
        q = time_events(lambda: ops.quad_solution_vector(b, r), dev, 6)
        out["quad_solution_vector_value"] = {"n": n, **q, "bytes_per_elt": 40,
                                             "GBps": 40.0 * n / (q["median_ms"] * 1e-3) / 1e9,
                                             "frac_of_hbm_peak": 40.0 * n / (q["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        q = time_events(lambda: ops.quad_solution_vector_derivs(b, r), dev, 6)
        out["quad_solution_vector_with_derivs"] = {"n": n, **q, "bytes_per_elt": 88,
                                                   "GBps": 88.0 * n / (q["median_ms"] * 1e-3) / 1e9,
                                                   "frac_of_hbm_peak": 88.0 * n / (q["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out["note"] = ("b uniform in [0, 1.3], r = 0.1: 85 % of the elements overlap the disk and evaluate the elliptic "
                   "integrals -- the op is fp64-VALU-bound there, not HBM-bound (SURVEY.md 8d caveat)")
    return out


def graphed(xo, fn, inputs, dev, iters):
    """time `fn(*inputs)` replayed as one hipGraph; eager launches if the capture fails"""
    try:
        g = xo.GraphedStep(fn, *inputs)
        q = time_events(lambda: g(), dev, iters)
        return q, "hipGraph replay"
    except Exception as exc:
        torch.cuda.synchronize(dev)
        q = time_events(lambda: fn(*inputs), dev, max(3, iters // 4))
        return q, f"eager launches (capture failed: {repr(exc)[:120]})"


def extra_c3(xo, leaves, t, dev, D):
    """BASELINE configs[2] (C3): the C2 light curve + a celerite SHO-term GP log-likelihood on the
    residual; value + gradient w.r.t. the orbit / limb-darkening leaves and the kernel hyper-parameters"""
    yobs = 5e-4 * torch.randn(N_CAD, dtype=torch.float64, device=dev)
    names = list(leaves)
    hyper = [torch.full((D,), v, dtype=torch.float64, device=dev, requires_grad=True) for v in (1e-3, 5.0, 0.7071)]

    def one(*vals):
        Lv = dict(zip(names, vals[:len(names)]))
        sigma, rho, Q = vals[len(names):]
        orbit = xo.KeplerianOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"])
        lc = xo.LimbDarkLightCurve(Lv["u1"], Lv["u2"]).get_light_curve(orbit=orbit, r=Lv["r"], t=t, total=True)
        gp = xo.gp.GaussianProcess(xo.gp.terms.SHOTerm(sigma=sigma, rho=rho, Q=Q), t=t, yerr=5e-4, mean=lc)
        ll = gp.log_likelihood(yobs)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    q, how = graphed(xo, one, list(leaves.values()) + hyper, dev, 12)
    J = 2
    bpu = 48 + 16 * (1 + J + J * J)
    gbps = bpu * D * N_CAD / (q["median_ms"] * 1e-3) / 1e9
    return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "draws": D, "launch": how,
            "survey_8d_bytes_per_unit": bpu, "survey_8d_GBps": gbps, "survey_8d_frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
            "note": "C3: C2 light curve + SHO-term celerite GP on the residual (recurrences in parallel over "
                    "time), value + gradient of every leaf incl. (sigma, rho, Q) per draw"}


def extra_c4(xo, dev, D=64):
    """BASELINE configs[3] (C4) at its per-GPU size: 4 planets, 200 000 cadences, 64 draws"""
    rng = np.random.default_rng(4)
    n = 200_000
    t = torch.arange(n, dtype=torch.float64, device=dev) * CADENCE
    base = dict(period=[3.5, 7.9, 13.1, 29.7], t0=[1.0, 2.3, 5.1, 11.7], b=[0.3, 0.1, 0.5, 0.2],
                ecc=[0.05, 0.1, 0.2, 0.3], omega=[1.1, -0.4, 2.0, 0.3], r=[0.1, 0.05, 0.07, 0.03])
    leaves = {}
    for k, v in base.items():
        x = np.asarray(v)[None, :] * (1 + 1e-3 * rng.normal(size=(D, 4)))
        if k == "ecc":
            x = np.clip(x, 0.0, 0.95)
        leaves[k] = torch.tensor(x, dtype=torch.float64, device=dev, requires_grad=True)
    leaves["u1"] = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
    leaves["u2"] = torch.full((D,), 0.2, dtype=torch.float64, device=dev, requires_grad=True)
    gbar = torch.randn(D, n, dtype=torch.float64, device=dev)
    names = list(leaves)
    from exoplanet_amd import ops

    def one(*vals):
        _, L, grads = step(xo, ops, dict(zip(names, vals)), t, gbar)
        return (L.detach(),) + grads

    q, how = graphed(xo, one, list(leaves.values()), dev, 40)
    gbps = SURVEY_BYTES_PER_UNIT * D * n / (q["median_ms"] * 1e-3) / 1e9
    return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "draws": D, "n_cadences": n, "n_planets": 4,
            "launch": how, "survey_8d_GBps": gbps, "survey_8d_frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
            "note": "C4 at the per-GPU size of its 8-GPU statement (512 draws / 8): dense summed flux, value + "
                    "gradient of all 4 x 6 + 2 leaves per draw"}


def extra_c5(xo, dev, D=128):
    """BASELINE configs[4] (C5) at its per-GPU size: 65 000 long cadences, exposure stencil x 7,
    secondary eclipse, three SHO terms (J = 6), 128 chains"""
    rng = np.random.default_rng(5)
    n = 65_000
    texp = 29.4 / 1440.0
    t = torch.arange(n, dtype=torch.float64, device=dev) * texp
    mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev,  # noqa: E731
                                requires_grad=True)
    leaves = {k: mk(v) for k, v in dict(period=2.7, t0=0.4, b=0.2, ecc=0.1, omega=0.7, r=0.08).items()}
    vec = lambda v: torch.full((D,), v, dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    leaves.update(sbr=vec(0.3), s1=vec(4e-4), s2=vec(3e-4), s3=vec(2e-4))
    yobs = 3e-4 * torch.randn(n, dtype=torch.float64, device=dev)
    ones = torch.ones(D, dtype=torch.float64, device=dev)
    names = list(leaves)
    T = xo.gp.terms

    def one(*vals):
        Lv = dict(zip(names, vals))
        orbit = xo.KeplerianOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"])
        lc = xo.SecondaryEclipseLightCurve((0.3, 0.2), (0.4, 0.1), Lv["sbr"]).get_light_curve(
            orbit=orbit, r=Lv["r"], t=t, texp=texp, oversample=7, total=True)
        kern = (T.SHOTerm(sigma=Lv["s1"], rho=20.0 * ones, Q=2.0 * ones) + T.SHOTerm(sigma=Lv["s2"], rho=10.0 * ones, Q=ones)
                + T.SHOTerm(sigma=Lv["s3"], rho=2.0 * ones, Q=0.7071 * ones))
        gp = xo.gp.GaussianProcess(kern, t=t, yerr=3e-4, mean=lc)
        ll = gp.log_likelihood(yobs)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    q, how = graphed(xo, one, list(leaves.values()), dev, 12)
    J = 6
    bpu = 48 + 16 * (1 + J + J * J)
    gbps = bpu * D * n / (q["median_ms"] * 1e-3) / 1e9
    return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "draws": D, "n_cadences": n, "launch": how,
            "survey_8d_bytes_per_unit": bpu, "survey_8d_GBps": gbps, "survey_8d_frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
            "note": "C5 at the per-GPU size of its 8-GPU statement (1024 chains / 8): secondary-eclipse light "
                    "curve (7 sub-exposures) + 3-term GP, value + gradient"}


def extra_astrometry(xo, dev, D=1024, n_epoch=64):
    """8f row 3: relative astrometry (separation, position angle) + star velocities of a two-planet system at a few
    dozen epochs, value + gradient of every orbit parameter: the fused position / velocity op against the composed
    torch path it replaces"""
    rng = np.random.default_rng(6)
    t = torch.tensor(np.sort(rng.uniform(0.0, 800.0, n_epoch)), dtype=torch.float64, device=dev)
    base = dict(period=[350.0, 97.0], t0=[10.0, 31.0], incl=[1.1, 1.3], ecc=[0.2, 0.5], omega=[0.3, -1.9], Omega=[0.4, 2.0],
                m_planet=[1e-3, 4e-4])
    leaves = {k: torch.tensor(np.asarray(v)[None, :] * (1 + 1e-3 * rng.normal(size=(D, 2))), dtype=torch.float64, device=dev,
                              requires_grad=True) for k, v in base.items()}
    names = list(leaves)
    w = [torch.randn(D, n_epoch, 2, dtype=torch.float64, device=dev) for _ in range(5)]

    def make(fused):
        def one(*vals):
            orbit = xo.KeplerianOrbit(m_star=1.0, r_star=1.0, **dict(zip(names, vals)))
            if not fused:
                orbit._fused_vector = lambda *a, **k: None
            rho, theta = orbit.get_relative_angles(t, parallax=0.05)
            vx, vy, vz = orbit.get_star_velocity(t)
            L = (rho * w[0] + theta * w[1] + vx * w[2] + vy * w[3] + vz * w[4]).sum((-1, -2))
            return (L.detach(),) + torch.autograd.grad(L.sum(), vals)
        return one

    out = {}
    for tag, fused in (("fused", True), ("composed", False)):
        q, how = graphed(xo, make(fused), list(leaves.values()), dev, 30)
        out[tag] = {"median_ms": q["median_ms"], "launch": how}
    out["draws"], out["n_epoch"], out["n_planet"] = D, n_epoch, 2
    out["note"] = ("astrometry + star velocities of 2 planets at %d epochs, %d draws, value + gradients of 7 x 2 leaves: "
                   "exo_orbit_vector_* (one launch each way per quantity) vs ops.kepler + torch algebra" % (n_epoch, D))
    return out


def extra_hmc(xo, ops, dev, D=1024, n_leapfrog=8, dense=False, nuts=False):
    """8f row 4: one HMC trajectory (n_leapfrog value + gradient evaluations of the C2 likelihood and the position /
    momentum updates between them) for D chains, replayed as one hipGraph"""
    rng = np.random.default_rng(8)
    t = torch.arange(N_CAD, dtype=torch.float64, device=dev) * CADENCE
    lv = make_leaves(D, 11, dev)
    names = [k for k in lv if k not in ("u1", "u2")]
    u1, u2 = lv["u1"].detach(), lv["u2"].detach()
    with torch.no_grad():
        one = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)  # noqa: E731
        orbit = xo.KeplerianOrbit(period=one(3.5), t0=one(1.0), b=one(0.3), ecc=one(0.3), omega=one(1.1))
        truth = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=one(0.1), t=t).sum(-1).reshape(-1)
    obs = truth + 1e-4 * torch.randn(N_CAD, dtype=torch.float64, device=dev)
    ivar = 1e8

    def logp(*vals):
        # Gaussian log-likelihood of the observed light curve (white noise): one call on the sparse light curve
        Lv = dict(zip(names, vals))
        orbit = xo.KeplerianOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"])
        if dense:
            flux = xo.LimbDarkLightCurve(u1, u2).get_light_curve(orbit=orbit, r=Lv["r"], t=t).sum(-1)
            return -0.5 * ivar * ((flux - obs) ** 2).sum(-1)
        return xo.LimbDarkLightCurve(u1, u2).white_noise_log_likelihood(orbit=orbit, r=Lv["r"], t=t, y=obs, yerr=ivar ** -0.5)

    params = [lv[k].detach().clone() for k in names]
    if nuts:
        # (unit masses and a step size far below the posterior's scales: every tree grows to the depth limit, i.e. the
        # leg times 2**4 - 1 = 15 leaves per transition and the tree bookkeeping around them, not a tuned sampler)
        smp = xo.NUTS(logp, params, step_size=1e-11, max_depth=4, generator=torch.Generator(device=dev).manual_seed(12))
        for _ in range(2):
            smp.step()
        torch.cuda.synchronize(dev)
        n0 = smp.n_leapfrog
        q = stats_loop(lambda _: smp.step(), dev, 10)
        leaves = (smp.n_leapfrog - n0) / 10.0
        return {"transitions_per_s": D / (q["median_ms"] * 1e-3), "evals_per_s": D * leaves / (q["median_ms"] * 1e-3), **q,
                "chains": D, "leaves_per_transition": leaves, "ms_per_leaf": q["median_ms"] / leaves,
                "mean_depth": float(smp.mean_depth().mean()),
                "note": "exoplanet_amd.NUTS: %d chains in lockstep, one batched value+gradient evaluation of the C2 white-noise "
                        "likelihood per leaf (leapfrog step replayed as a hipGraph), tree bookkeeping (multinomial sampling, "
                        "checkpointed turning checks, masks) as torch ops on the device, one host synchronisation per doubling" % D}
    hmc = xo.HMC(logp, params, step_size=1e-6, n_leapfrog=n_leapfrog)
    for _ in range(3):
        hmc.step()
    torch.cuda.synchronize(dev)
    q = stats_loop(lambda _: hmc.step(), dev, 30)
    evals = D * (n_leapfrog + 1)
    return {"trajectories_per_s": D / (q["median_ms"] * 1e-3), "evals_per_s": evals / (q["median_ms"] * 1e-3), **q,
            "chains": D, "n_leapfrog": n_leapfrog,
            "note": "exoplanet_amd.HMC: %d chains, %d leapfrog steps per trajectory = %d value+gradient evaluations of the "
                    "C2 white-noise likelihood (LimbDarkLightCurve.white_noise_log_likelihood: exo_transit_chi2_vjp_f64, one "
                    "evaluation per solved cadence; `dense_ms`: the same through get_light_curve and torch passes over (chains, "
                    "cadences) arrays), one hipGraph replay per trajectory + momentum draw and accept / reject on the device"
                    % (D, n_leapfrog, n_leapfrog + 1)}


def extra_ttv(xo, ops, leaves, t, gbar, dev, D):
    n_tr = int((float(t[-1]) - 1.0) / 3.5) + 1
    offs = torch.tensor(0.01 * np.random.default_rng(7).normal(size=(D, n_tr)), dtype=torch.float64, device=dev,
                        requires_grad=True)
    tnames = list(leaves) + ["ttvs"]

    def ttv_step(*vals):
        Lv = dict(zip(tnames, vals))
        orb = xo.orbits.TTVOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"],
                                 ttvs=[Lv["ttvs"]])
        rec_t, ld_t, _, fl = orb.kernel_inputs(Lv["r"], (Lv["u1"], Lv["u2"]))
        ed, sh = orb.kernel_ttv()
        _, dot = ops.transit_flux_dot(t, rec_t, ld_t, gbar, flags=fl, ttv=(ed.contiguous(), sh.contiguous()))
        return (dot.detach(),) + torch.autograd.grad(dot.sum(), vals)

    q, how = graphed(xo, ttv_step, list(leaves.values()) + [offs], dev, 30)
    # the same orbit as a white-noise likelihood (exo_transit_chi2_ttv_vjp_f64): no (draw, cadence) array
    obs = 1e-4 * torch.randn(t.numel(), dtype=torch.float64, device=dev)

    def ttv_like(*vals):
        Lv = dict(zip(tnames, vals))
        orb = xo.orbits.TTVOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"],
                                 ttvs=[Lv["ttvs"]])
        ll = xo.LimbDarkLightCurve(Lv["u1"], Lv["u2"]).white_noise_log_likelihood(orbit=orb, r=Lv["r"], t=t, y=obs, yerr=1e-4)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    ql, _ = graphed(xo, ttv_like, list(leaves.values()) + [offs], dev, 30)
    return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "transits": n_tr, "launch": how,
            "likelihood": {"evals_per_s": D / (ql["median_ms"] * 1e-3), "median_ms": ql["median_ms"]},
            "note": "C2 step with a TTVOrbit (timing tables in the fused kernels, gradients to every per-transit offset); "
                    "`likelihood`: the same orbit through white_noise_log_likelihood (one evaluation per solved cadence)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--draws-per-gpu", type=int, default=1024)
    ap.add_argument("--global-draws", type=int, default=0,
                    help="fix the TOTAL number of draws (strong scaling: BASELINE C4 = 512, C5 = 1024 over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="skip the >= 100 steps / >= 2 s median / p10 / p90 loop")
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
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod

    import exoplanet_amd as xo
    from exoplanet_amd import ops, _lib
    from exoplanet_amd.distributed import LoglikeExchange, shard_bounds

    _lib.load()  # fail loudly if the HIP library is missing
    if args.global_draws:
        lo, hi = shard_bounds(args.global_draws, rank, world)
        D, n_global, scaling = hi - lo, args.global_draws, "strong"
    else:
        D, n_global, scaling = args.draws_per_gpu, world * args.draws_per_gpu, "weak"
    t = torch.arange(N_CAD, dtype=torch.float64, device=dev) * CADENCE
    gbar = torch.as_tensor(np.random.default_rng(2 + rank).normal(size=(D, N_CAD)), device=dev)
    leaves = make_leaves(D, 100 + rank, dev)
    exchange = LoglikeExchange(n_global, dev) if dist is not None else None
    events = HipEvents(max(args.steps, 20))

    def one(i, ev=True):
        evs = events.handles(i) if (ev and i >= 0) else (None, None)
        flux, L, grads = step(xo, ops, leaves, t, gbar, events=evs)
        if exchange is not None:
            exchange_step(exchange, L, pipelined=True)
        return flux, L, grads

    # The step is a dozen short launches (packing kernel, window, scan, heavy, reduce, packing VJP and
    # a few tensor-shuffling torch kernels): launch-bound when issued eagerly, so the timed region
    # replays it as ONE hipGraph; the collective -- exactly one call, straight from the graph's static
    # output -- stays outside the graph.
    graph = None
    static = {}
    if not args.no_graph:
        names = list(leaves)
        graph = xo.GraphedStep(lambda *vals: step(xo, ops, dict(zip(names, vals)), t, gbar), *leaves.values())
        static["flux"], static["L"], static["grads"] = graph.outputs

    def one_graph(i):
        graph()
        if exchange is not None:
            exchange_step(exchange, static["L"], pipelined=True)

    run = one_graph if graph is not None else (lambda i: one(i, ev=False))
    # setup, outside the contract's W + K steps: bring clocks, caches and the allocator to the state a sampler runs
    # in (a few hundred steps, ~0.1 s; the same count on every rank)
    SETUP_STEPS = 300
    for _ in range(SETUP_STEPS):
        run(-1)
    torch.cuda.synchronize(dev)
    drain = exchange.finish if exchange is not None else None
    wall = time_steps(run, args.steps, args.warmup, dist, dev, drain=drain)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    timing = None
    if not args.no_stats:
        # >= 100 steps and >= 2 s, whatever --steps was; the count follows from the
        # max-over-ranks step time above, so it is the same on every rank
        n_stat = int(min(6000, max(100, np.ceil(2.0 / max(wall / args.steps, 1e-6)))))
        timing = stats_loop(run, dev, n_stat)
        if exchange is not None:
            exchange.finish()
        if dist is not None:
            dist.barrier()
    # hipEvents cannot bracket a node inside a replayed graph: time the kernels of the sweep over
    # directly launched steps on the same inputs (events recorded by the C ABI on the launch stream)
    n_ev = len(events.pairs)
    for i in range(n_ev):
        one(i)
    torch.cuda.synchronize(dev)
    kernel_ms, per_launch = events.mean_ms()
    # cadences the sweep solves = the runs of the conjunction windows (one sparse sweep, outside any timing)
    with torch.no_grad():
        orbit0 = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                                   omega=leaves["omega"])
        rec0, ld0, _, flags0 = orbit0.kernel_inputs(leaves["r"], (leaves["u1"], leaves["u2"]), use_in_transit=False)
        n_active = ops.transit_flux_sparse(t, rec0.detach(), ld0.detach(), flags=flags0).n_solved()

    out = None
    if rank == 0:
        evals = n_global * args.steps
        # bytes this design must move per sweep: the dense flux array once (zeros); for every solved
        # cadence t and gbar in, its flux out to the run-ordered value array, back in and out to its
        # place in the flux array (the binary searches of the windows are a few MB)
        req = {"flux_zero_fill": 8 * D * N_CAD, "t_and_gbar_solved": 16 * n_active,
               "value_and_cadence_arrays_write_read": 24 * n_active, "flux_write_solved": 8 * n_active}
        req_bytes = sum(req.values())
        achieved = req_bytes / (kernel_ms * 1e-3) / 1e9
        survey_bytes = SURVEY_BYTES_PER_UNIT * D * N_CAD
        traffic = measured_traffic(D)
        out = {
            "metric": "light-curve evals/sec (value+grad) at 150k cadences",
            "value": evals / wall,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "setup_steps_before_warmup": SETUP_STEPS,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] (C2): single planet e=0.3 Kepler solve + quadratic limb-darkened "
                            "transit, 150000 cadences, value+grad, use_in_transit=False: dense flux output, every "
                            f"cadence classified on the device, {100.0 * n_active / (D * N_CAD):.2f} % solved",
                "n_cadences": N_CAD, "draws_per_gpu": D, "global_draws": n_global,
                "parallelism": f"draws sharded over {world} GPU(s); one collective of per-draw scalars per step, issued "
                               "asynchronously from a private copy (double-buffered): it overlaps the next step's kernels and "
                               "is waited for inside the timed region",
                "step": "leaf params (separate tensors, read in place) -> record-packing kernel (orbit algebra + "
                        "get_cl) -> window + run-enumeration + heavy kernels (value+VJP, one sweep; the heavy kernel's blocks "
                        "finish their own draws at >= 512 draws, a separate finish kernel below that) -> packing "
                        "VJP kernel (cotangent of L folded in) -> leaf gradients: five launches (six below 512 draws)"
                        + ("; replayed as one hipGraph" if graph is not None else "; eager launches"),
            },
            "timing": timing,
            "roofline": {
                "bound": "hbm",
                "kernel": "transit_window_kernel (+ sortedness flags) + transit_enum_kernel + transit_runs_kernel "
                          "[+ transit_finish_kernel below 512 draws] (one sweep)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "algorithmic_bytes_per_launch": req_bytes, "algorithmic_bytes_breakdown": req,
                "active_cadences_per_launch": n_active, "kernel_ms": kernel_ms,
                "kernel_ms_quantiles": quantiles(per_launch),
                "survey_8d_count": {"bytes_per_unit": SURVEY_BYTES_PER_UNIT, "bytes_per_launch": survey_bytes,
                                    "GBps": survey_bytes / (kernel_ms * 1e-3) / 1e9,
                                    "note": "SURVEY.md 8d charges t + gbar + flux = 24 B to EVERY (draw, cadence); "
                                            "this design finds the cadences to solve by binary search and reads t, gbar "
                                            "only there, so this figure is not a fraction of anything: `frac` above is "
                                            "against the bytes the design must move"},
                "note": "achieved = algorithmic_bytes_per_launch / mean hipEvent time of the launches of one sweep "
                        "(sortedness flags, window constants, run enumeration, heavy = solved cadences + zero-fill of "
                        "the dense flux, then -- in the same kernel when a draw is one block's work, else in a last small one -- "
                        "values to their cadences + block partials), eager launches on the "
                        "same inputs as the timed graph.  The heavy kernel is where the time goes: fp64 VALU issue "
                        "(~1e3 flop per solved cadence) with the store stream of the dense output interleaved; "
                        "rocprof per-kernel averages and PMC traffic: profiles/",
            },
        }

    # the extra legs are single-GPU diagnostics: under torch.distributed.run they would only add
    # barriers that every rank has to reach (an exception on one rank would hang the others)
    if not args.no_extras and world == 1 and dist is None:
        extras = {}

        def leg(name, fn):
            try:
                extras[name] = fn()
            except Exception as exc:  # an extra leg must not take the headline measurement down with it
                extras[name] = {"error": repr(exc)[:300]}
            torch.cuda.synchronize(dev)

        def in_transit():
            names = list(leaves)
            q, how = graphed(xo, lambda *v: step(xo, ops, dict(zip(names, v)), t, gbar, use_in_transit=True),
                             list(leaves.values()), dev, 50)
            return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "launch": how,
                    "note": "the reference's default use_in_transit=True: contact-point windows from the packing "
                            "kernel decide what is solved"}

        def op_level():
            orbit = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                                      omega=leaves["omega"])
            rec, c, _, _ = orbit.kernel_inputs(leaves["r"], (leaves["u1"], leaves["u2"]))
            rec, c = rec.detach(), c.detach()
            q = time_events(lambda: ops.transit_flux_value_and_vjp(t, rec, c, gbar), dev, 50)
            return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q,
                    "note": "fused kernel call only (4 launches), no orbit algebra / autograd"}

        def small_batch():
            res = {}
            for d in (64, 128, 256):
                lv = make_leaves(d, 300 + d, dev)
                gb = gbar[:d].contiguous()
                nm = list(lv)
                q, how = graphed(xo, lambda *v: step(xo, ops, dict(zip(nm, v)), t, gb), list(lv.values()), dev, 50)
                res[str(d)] = {"evals_per_s": d / (q["median_ms"] * 1e-3), "median_ms": q["median_ms"], "launch": how}
            return res

        def sparse_output():
            names = list(leaves)

            def sp_step(*vals):
                _, L, grads = step(xo, ops, dict(zip(names, vals)), t, gbar, sparse=True)
                return (L.detach(),) + grads

            q, how = graphed(xo, sp_step, list(leaves.values()), dev, 50)
            return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "launch": how,
                    "note": "the same C2 step with EXO_FLAG_SPARSE: the output is the runs of cadences in which the planet "
                            "can overlap the disk plus their flux values (every other cadence is exactly 0) -- no 1.23 GB "
                            "of zeros; what a likelihood needs"}

        def light_delay():
            names = list(leaves)

            def ld_step(*vals):
                _, L, grads = step(xo, ops, dict(zip(names, vals)), t, gbar, light_delay=True)
                return (L.detach(),) + grads

            q, how = graphed(xo, ld_step, list(leaves.values()), dev, 50)
            return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "launch": how,
                    "note": "the C2 step with light_delay=True (keplerian.py:411-470): every solved sample is evaluated at "
                            "its retarded time -- a second Kepler solve and the reverse sweep through the delay, in the "
                            "same kernel; dense output"}

        def likelihood():
            names = list(leaves)
            obs = 1e-4 * torch.randn(N_CAD, dtype=torch.float64, device=dev)

            def ll_step(*vals):
                Lv = dict(zip(names, vals))
                orbit = xo.KeplerianOrbit(period=Lv["period"], t0=Lv["t0"], b=Lv["b"], ecc=Lv["ecc"], omega=Lv["omega"])
                ll = xo.LimbDarkLightCurve(Lv["u1"], Lv["u2"]).white_noise_log_likelihood(orbit=orbit, r=Lv["r"], t=t, y=obs, yerr=1e-4)
                return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

            q, how = graphed(xo, ll_step, list(leaves.values()), dev, 50)
            return {"evals_per_s": D / (q["median_ms"] * 1e-3), **q, "launch": how,
                    "note": "C2 as a white-noise likelihood, value + gradient of every leaf (exo_transit_chi2_vjp_f64: one planet, "
                            "one sample per cadence -> ONE evaluation per solved cadence, the cotangent 2 w (F - obs) formed "
                            "inside it; no (draw, cadence) array exists)"}

        leg("c2_white_noise_likelihood", likelihood)
        leg("c2_sparse_output", sparse_output)
        leg("c2_light_delay", light_delay)
        leg("in_transit_only", in_transit)
        leg("op_level_every_cadence", op_level)
        leg("c2_small_batches", small_batch)
        leg("c2_with_transit_timing_variations", lambda: extra_ttv(xo, ops, leaves, t, gbar, dev, D))
        leg("c3_light_curve_plus_sho_gp", lambda: extra_c3(xo, leaves, t, dev, D))
        del gbar
        torch.cuda.empty_cache()
        leg("c4_four_planets_64_draws", lambda: extra_c4(xo, dev))
        leg("c5_secondary_eclipse_3term_gp_128_chains", lambda: extra_c5(xo, dev))
        torch.cuda.empty_cache()
        leg("astrometry_and_velocities", lambda: extra_astrometry(xo, dev))
        def nuts_leg():
            return extra_hmc(xo, ops, dev, nuts=True)

        def hmc_leg():
            res = extra_hmc(xo, ops, dev)
            res["dense_ms"] = extra_hmc(xo, ops, dev, dense=True)["median_ms"]
            return res

        leg("hmc_trajectory_c2", hmc_leg)
        leg("nuts_transition_c2", nuts_leg)
        torch.cuda.empty_cache()
        leg("ops", lambda: extra_ops(ops, dev))
        if rank == 0:
            out["extras"] = extras

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
