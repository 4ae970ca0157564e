"""Device-resident batched Hamiltonian Monte Carlo on top of the value + gradient hot path.

The reference hands its log-likelihood graph to PyMC (`pm.sample`,
/root/reference/docs/tutorials/data-and-models.md:289,402,507), which runs one chain per *process*
(/root/reference/docs/user/multiprocessing.rst:6-8).  Here the D chains of a batch are the draw
dimension of the kernels: positions, momenta and the per-chain accept / reject decision live on the
GPU, a whole leapfrog trajectory -- L value+gradient evaluations and the position / momentum updates
between them -- is captured ONCE as a hipGraph and replayed, and nothing synchronises with the host
inside a step.  HMC with a fixed trajectory length and a diagonal mass matrix: the driver that turns "value +
gradient provider" into an end-to-end sampler step.  `HMC.warmup` adapts what a fixed-length
sampler can adapt without leaving the device: one step size PER CHAIN by dual averaging (Hoffman & Gelman 2014,
section 3.2) and, optionally, the diagonal mass matrix from the variance pooled over chains and warm-up draws.
The step sizes and masses are device tensors read by the captured trajectory, so adapting them needs no re-capture.
"""
import torch

from .graph import GraphedStep

__all__ = ["HMC"]


class HMC:
    """``logp_fn(*params) -> (D,)`` log-density of every chain; ``params``: tensors with a leading
    chain dimension D (their values are the chains' starting points and are updated in place).

    ``step()`` runs one trajectory for all chains and returns the (D,) bool accept mask (a device
    tensor: reading it is the caller's choice of when to synchronise).  ``mass``: one tensor per
    parameter (broadcastable to it), default 1.  On a ROCm device the trajectory is replayed as a
    hipGraph (``graph=False`` launches it eagerly); on the CPU it simply runs -- the sampler logic is
    device-agnostic, which is how tests/test_sampling.py checks it without a GPU.
    """

    def __init__(self, logp_fn, params, step_size, n_leapfrog, mass=None, graph=True, generator=None):
        self.params = [p.detach() for p in params]
        if not self.params or any(p.shape[0] != self.params[0].shape[0] for p in self.params):
            raise ValueError("params must be tensors with a common leading chain dimension")
        self.logp_fn = logp_fn
        self.L = int(n_leapfrog)
        if self.L < 1 or not float(step_size) > 0:
            raise ValueError("need step_size > 0 and n_leapfrog >= 1")
        # one step size per chain, a device tensor (the captured trajectory reads it: adapting it is an in-place update)
        self.eps = torch.full((self.params[0].shape[0],), float(step_size), dtype=self.params[0].dtype,
                              device=self.params[0].device)
        self.mass = [torch.ones_like(p) if m is None else torch.as_tensor(m, dtype=p.dtype, device=p.device).expand_as(p).clone()
                     for p, m in zip(self.params, mass or [None] * len(self.params))]
        self.generator = generator
        self.D = self.params[0].shape[0]
        self.n_steps = 0
        self.n_accept = torch.zeros(self.D, dtype=torch.float64, device=self.params[0].device)
        self._graph = None
        if graph and self.params[0].is_cuda:
            k = len(self.params)
            # static inputs: positions, momenta; static outputs: proposal, its momenta, both log-densities
            q0 = [p.clone() for p in self.params]
            p0 = [torch.zeros_like(p) for p in self.params]
            self._graph = GraphedStep(lambda *a: self._trajectory(list(a[:k]), list(a[k:])), *q0, *p0)

    # one value + gradient evaluation of all chains
    def _value_and_grad(self, q):
        with torch.enable_grad():
            qs = [x.detach().requires_grad_(True) for x in q]
            lp = self.logp_fn(*qs)
            grads = torch.autograd.grad(lp.sum(), qs)
        return lp.detach(), [g.detach() for g in grads]

    def _trajectory(self, q, p):
        """leapfrog: (q, p) -> (q', p', logp(q), logp(q'))"""
        lp0, g = self._value_and_grad(q)
        q = [x.clone() for x in q]
        p = [x.clone() for x in p]
        lp = lp0
        eps = [self.eps.reshape((self.D,) + (1,) * (x.dim() - 1)) for x in q]
        for _ in range(self.L):
            p = [pi + 0.5 * ei * gi for pi, gi, ei in zip(p, g, eps)]
            q = [qi + ei * pi / mi for qi, pi, mi, ei in zip(q, p, self.mass, eps)]
            lp, g = self._value_and_grad(q)
            p = [pi + 0.5 * ei * gi for pi, gi, ei in zip(p, g, eps)]
        return tuple(q) + tuple(p) + (lp0, lp)

    def _kinetic(self, p):
        return sum((0.5 * pi * pi / mi).reshape(self.D, -1).sum(-1) for pi, mi in zip(p, self.mass))

    @torch.no_grad()
    def step(self):
        k = len(self.params)
        p0 = [torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=self.generator) * torch.sqrt(m)
              for q, m in zip(self.params, self.mass)]
        if self._graph is not None:
            out = self._graph(*self.params, *p0)
        else:
            out = self._trajectory(self.params, p0)
        q1, p1, lp0, lp1 = list(out[:k]), list(out[k:2 * k]), out[2 * k], out[2 * k + 1]
        dH = (lp1 - self._kinetic(p1)) - (lp0 - self._kinetic(p0))       # -(H1 - H0)
        u = torch.rand(self.D, dtype=dH.dtype, device=dH.device, generator=self.generator)
        accept = torch.log(u) < dH                                        # NaN / -inf proposals are rejected
        for q, qn in zip(self.params, q1):
            m = accept.reshape((self.D,) + (1,) * (q.dim() - 1))
            q.copy_(torch.where(m, qn, q))
        self.n_steps += 1
        self.n_accept += accept.to(self.n_accept.dtype)
        self.last_logp = torch.where(accept, lp1, lp0)
        # acceptance probability min(1, exp(-dH)) of every chain (NaN proposals: 0): what step-size adaptation feeds on
        self.last_accept_prob = torch.nan_to_num(torch.exp(torch.clamp(dH, max=0.0)), nan=0.0)
        return accept

    @torch.no_grad()
    def warmup(self, n_steps, target_accept=0.8, adapt_mass=False, gamma=0.05, t0=10.0, kappa=0.75):
        """``n_steps`` trajectories that adapt, per chain, the step size to the acceptance rate ``target_accept``
        by dual averaging; afterwards the averaged step sizes are kept.  ``adapt_mass``: half way through, the
        diagonal masses become 1 / variance of every parameter element, pooled over chains and the draws so far (the
        chains of a batch target one posterior), and the step-size adaptation restarts.  Nothing here synchronises
        with the host; statistics of the sampling phase (``accept_rate``) start after the warm-up."""
        def restart():
            mu = torch.log(10.0 * self.eps)
            return mu, torch.zeros_like(self.eps), torch.zeros_like(self.eps)   # mu, Hbar, log eps_bar

        mu, Hbar, log_eps_bar = restart()
        m = 0
        half = n_steps // 2 if adapt_mass else -1
        s1 = [torch.zeros_like(q[0]) for q in self.params]
        s2 = [torch.zeros_like(q[0]) for q in self.params]
        n_acc = 0
        for it in range(n_steps):
            self.step()
            m += 1
            Hbar = (1.0 - 1.0 / (m + t0)) * Hbar + (target_accept - self.last_accept_prob) / (m + t0)
            log_eps = mu - (m ** 0.5 / gamma) * Hbar
            w = m ** (-kappa)
            log_eps_bar = w * log_eps + (1.0 - w) * log_eps_bar
            self.eps.copy_(torch.exp(log_eps))
            if adapt_mass and it < half:
                for a, b, q in zip(s1, s2, self.params):
                    a += q.sum(0); b += (q * q).sum(0)
                n_acc += self.D
            if it + 1 == half and n_acc > 1:
                for mass, a, b in zip(self.mass, s1, s2):
                    var = (b - a * a / n_acc) / (n_acc - 1)
                    ok = torch.isfinite(var) & (var > 0)
                    mass.copy_(torch.where(ok, 1.0 / var, mass[0]).expand_as(mass))
                self.eps.copy_(torch.exp(log_eps_bar))
                mu, Hbar, log_eps_bar = restart()
                m = 0
        self.eps.copy_(torch.exp(log_eps_bar))
        self.n_steps = 0
        self.n_accept.zero_()
        return self.eps

    def accept_rate(self):
        """per-chain acceptance fraction so far (device tensor)"""
        return self.n_accept / max(self.n_steps, 1)
