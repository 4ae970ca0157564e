"""CPU: the conjunction-window bound of the light-curve sweep (transit_window_kernel, restated in
oracle/numpy_port.py::conjunction_window) against brute force over the true anomaly: inside the first bound
asin((1 + r) / ((a/R)(1 - e))) no overlap of the disks lies outside the refined window -- for transits and
occultations, either sign of sin i, e up to 0.99, a/R from 1.5 to 500, r from 1e-3 to 1, grazing and non-transiting
geometries included -- and the window is tight.  (The kernel itself is checked on the GPU against the fp64 scan of
every cadence: tests/test_gpu_scan_filter.py, tests/test_gpu_runs.py.)"""
import numpy as np

from oracle import numpy_port as P


def _overlap_range(e, w, ci, si, aor, ror, f0, d0, event, n=40001):
    f = np.linspace(f0 - d0, f0 + d0, n)
    dist = aor * (1 - e * e) / (1 + e * np.cos(f))
    th = w + f
    sep2 = dist * dist * (np.cos(th) ** 2 + ci * ci * np.sin(th) ** 2)
    z = dist * np.sin(th) * si                       # > 0: the body is on the observer's side
    front = z > 0 if event == 0 else z < 0
    hit = f[(sep2 < (1 + ror) ** 2) & front]
    return (hit.min(), hit.max()) if hit.size else None


def test_window_holds_every_overlap_and_is_tight():
    rng = np.random.default_rng(7)
    ratios, n_hit, n_miss = [], 0, 0
    for _ in range(6000):
        e = rng.uniform(0, 0.99) if rng.uniform() < 0.8 else 0.0
        w = rng.uniform(-np.pi, np.pi)
        aor = 10 ** rng.uniform(np.log10(1.5), np.log10(500))
        ror = 10 ** rng.uniform(-3, 0)
        b = rng.uniform(0, 1.5 + ror)
        event = int(rng.integers(0, 2))
        ci = b * (1 + e * np.sin(w)) / (1 - e * e) / aor          # keplerian.py:212-219
        if abs(ci) >= 1:
            continue
        si = np.sqrt(1 - ci * ci) * (1 if rng.uniform() < 0.8 else -1)
        win = P.conjunction_window(e, w, ci, si, aor, ror, event)
        if win is None:
            continue
        f0, d_lo, d_hi = win
        d0 = np.arcsin((1 + ror) / (aor * (1 - e)))
        assert d_lo <= d0 * (1 + 1e-6) + 1e-6 + 1e-15 and d_hi <= d0 * (1 + 1e-6) + 1e-6 + 1e-15
        rng_f = _overlap_range(e, w, ci, si, aor, ror, f0, d0, event)
        if rng_f is None:
            n_miss += 1
            continue
        n_hit += 1
        assert rng_f[0] >= f0 - d_lo and rng_f[1] <= f0 + d_hi, (e, w, aor, ror, b, event)
        ratios.append((d_lo + d_hi) / max(rng_f[1] - rng_f[0], 1e-9))
    assert n_hit > 1500 and n_miss > 200
    assert np.median(ratios) < 1.002 and np.percentile(ratios, 90) < 1.02


def test_window_of_the_benchmark_orbit():
    """C2 (e = 0.3, omega = 1.1, b = 0.3, a/R = 9.70, r = 0.1): the first bound is 6.8 % wider than the contacts, the
    refined window less than 0.1 %; one round per side is not enough on the far side"""
    e, w, b, aor, ror = 0.3, 1.1, 0.3, 9.70, 0.1
    ci = b * (1 + e * np.sin(w)) / (1 - e * e) / aor
    si = np.sqrt(1 - ci * ci)
    f0, d_lo, d_hi = P.conjunction_window(e, w, ci, si, aor, ror)
    d0 = np.arcsin((1 + ror) / (aor * (1 - e)))
    lo, hi = _overlap_range(e, w, ci, si, aor, ror, f0, d0, 0, n=2_000_001)
    exact = hi - lo
    assert 1.06 < 2 * d0 / exact < 1.075
    assert 1.0 <= (d_lo + d_hi) / exact < 1.001
    _, a1, b1 = P.conjunction_window(e, w, ci, si, aor, ror, rounds=1)
    assert (a1 + b1) / exact > (d_lo + d_hi) / exact
    # a planet that never reaches the disk keeps only the safety margin
    ci2 = 1.3 * (1 + e * np.sin(w)) / (1 - e * e) / aor
    _, a2, b2 = P.conjunction_window(e, w, ci2, np.sqrt(1 - ci2 * ci2), aor, ror)
    assert a2 < 2e-6 and b2 < 2e-6
