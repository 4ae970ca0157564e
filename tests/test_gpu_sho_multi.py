"""A sum of SHO terms in one launch each way (exo_sho_coefficients_multi_f64) against the per-term op: the same bits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _terms(n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    out = []
    from exoplanet_amd import ops

    for k, fl in enumerate((0, ops.SHO_SIGMA | ops.SHO_RHO, ops.SHO_SIGMA | ops.SHO_RHO | ops.SHO_TAU, ops.SHO_TAU)):
        amp = (0.1 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
        freq = (0.5 + 3.0 * torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
        # both regimes, and draws sitting on the clamp around Q = 1/2
        damp = (0.2 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
        if not fl & ops.SHO_TAU:
            damp[::7] = 0.5
            damp[1::7] = 0.5 + 1e-7
        out.append((amp, freq, damp, fl))
    return out


@pytest.mark.parametrize("n", [1, 63, 1000])
def test_multi_equals_per_term(n):
    from exoplanet_amd import ops

    dev = torch.device("cuda:0")
    terms = _terms(n, 5 + n, dev)
    leaves = [[x.clone().requires_grad_(True) for x in t[:3]] for t in terms]
    coef, kind = ops.sho_coefficients_multi([(a, f, d, t[3]) for (a, f, d), t in zip(leaves, terms)], 1e-5)
    assert coef.shape == (n, 4, 4) and kind.shape == (n, 4) and kind.dtype == torch.int32
    g = torch.randn(n, 4, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(dev)
    (coef * g).sum().backward()
    for k, t in enumerate(terms):
        one = [x.clone().requires_grad_(True) for x in t[:3]]
        c1, k1 = ops.sho_coefficients(*one, t[3], 1e-5)
        assert torch.equal(c1, coef[:, k]) and torch.equal(k1, kind[:, k])
        (c1 * g[:, k]).sum().backward()
        for a, b in zip(one, leaves[k]):
            assert torch.equal(a.grad, b.grad)
    assert set(kind.unique().tolist()) <= {0, 1}


def test_term_sum_takes_the_fused_op_and_the_loglike_agrees():
    from exoplanet_amd import ops
    from exoplanet_amd.gp import GaussianProcess, terms

    dev = torch.device("cuda:0")
    D, N = 8, 300
    g = torch.Generator().manual_seed(3)
    t = torch.sort(torch.rand(N, generator=g, dtype=torch.float64) * 20)[0].to(dev)
    y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
    par = [(0.3 + torch.rand(D, generator=g, dtype=torch.float64)).to(dev) for _ in range(6)]

    def ll(fused):
        ps = [p.clone().requires_grad_(True) for p in par]
        k = terms.SHOTerm(sigma=ps[0], rho=2.0 + ps[1], tau=1.0 + ps[2]) + terms.SHOTerm(sigma=ps[3], rho=5.0 + ps[4], Q=0.2 + ps[5])
        if not fused:
            k._fused_sho = lambda: None
        gp = GaussianProcess(k, t=t, diag=torch.full((N,), 0.01, dtype=torch.float64, device=dev))
        out = gp.log_likelihood(y)
        out.sum().backward()
        return out.detach(), [p.grad for p in ps]

    a, ga = ll(True)
    b, gb = ll(False)
    assert a.shape == (D,) and torch.equal(a, b)
    for x, z in zip(ga, gb):
        assert torch.equal(x, z)
    # a scalar sum (no batch) goes the same way
    k = terms.SHOTerm(sigma=par[0][0], rho=3.0, tau=2.0) + terms.SHOTerm(sigma=par[3][0], rho=7.0, Q=0.3)
    e, _, c, kd = k.pair_coefficients()
    assert e.shape == (0,) and c.shape == (2, 4) and kd.shape == (2,)


def test_bad_arguments():
    from exoplanet_amd import ops

    dev = torch.device("cuda:0")
    x = torch.ones(4, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        ops.sho_coefficients_multi([(x, x, x[:3], 0)])
    with pytest.raises(ValueError):
        ops.sho_coefficients_multi([(x, x, x, 0)] * 9)
    with pytest.raises(RuntimeError):
        ops.sho_coefficients_multi([(x, x, x, 64)])
