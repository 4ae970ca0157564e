"""GPU: a dense flux array kept across steps (ops.KeptDenseFlux, exo_transit_sparse_scatter_f64) -- the cadences the last step
solved zeroed, the sparse sweep, the cadences this step solved written -- against the dense sweep, step after step, bit for bit.
reference: src/exoplanet/light_curves/limb_dark.py:163-170, 228-230 (the dense light curve a sampler asks for at every step)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_runs import system

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


@pytest.mark.parametrize("planets,secondary,texp", [(1, False, False), (3, False, True), (1, True, True)])
def test_kept_dense_equals_the_dense_sweep_step_after_step(dev, planets, secondary, texp):
    from exoplanet_amd import ops

    rng = np.random.default_rng(5 + planets)
    D, N = 37, 20_011
    t = ops.vouch_sorted(T(np.arange(N) * (2.0 / 1440.0) + 0.25, dev))
    gflux = T(rng.normal(size=(D, N)), dev)
    flags = ops.FLAG_SECONDARY if secondary else 0
    kw = {}
    if texp:
        dt, w = P.exposure_stencil(5, 1)
        kw = dict(texp=T([0.02], dev), stencil_dt=T(dt, dev), stencil_w=T(w, dev))
    rec0, c = system(rng, D, planets, secondary)
    keeper = ops.KeptDenseFlux(t, D, planets, flags=flags, **kw)
    assert float(keeper.flux.abs().max()) == 0.0
    for step in range(5):
        rec = rec0.copy()
        rec[:, :, P.P_TP] += 0.06 * step * (1 + 0.2 * rng.normal(size=rec.shape[:2]))   # the transits move by more than their width
        rec[:, :, P.P_ROR] *= 1 + 0.02 * rng.normal(size=rec.shape[:2])
        params, ld = T(rec, dev), T(c, dev)
        want = ops.transit_flux_value_and_vjp(t, params, ld, gflux, flags=flags, **kw)
        got = keeper.step(params, ld, gflux)
        assert (want[0] < -1e-4).sum() > 20 * D
        assert torch.equal(got[0], want[0]), step
        for a, b in zip(got[1:3], want[1:3]):
            assert torch.equal(a, b)
    f = keeper.step(params, ld)                    # value only
    assert torch.equal(f, ops.transit_flux(t, params, ld, flags=flags, **kw))
    with pytest.raises(ValueError):
        ops.KeptDenseFlux(t, D, planets, flags=ops.FLAG_PER_PLANET)
