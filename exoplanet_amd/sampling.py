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

`NUTS` is the No-U-Turn sampler (Hoffman & Gelman 2014; multinomial variant with the generalised turning
criterion of Betancourt 2017, as in Stan / PyMC -- what `pm.sample` runs for the reference's models) over the same
batch of chains: every chain grows its own trajectory by doubling, all chains in lockstep leaf by leaf (one batched
value + gradient evaluation per leaf, chains whose tree has ended are masked), the turning checks of the sub-trees
through O(depth) momentum checkpoints (the iterative tree building of Phan, Pradhan & Jankowiak 2019), one host
synchronisation per DOUBLING ("is any chain still growing?") instead of one per leaf.
"""
import torch

from .graph import GraphedStep

__all__ = ["HMC", "NUTS"]


class _ChainSampler:
    """what HMC and NUTS share: parameter / mass bookkeeping, one value + gradient evaluation of all chains, warm-up"""

    def _setup(self, logp_fn, params, step_size, mass, generator):
        self.params = [p.detach() for p in params]
        if not self.params or any(p.shape[0] != self.params[0].shape[0] for p in self.params):
            raise ValueError("params must be tensors with a common leading chain dimension")
        if not float(step_size) > 0:
            raise ValueError("need step_size > 0")
        self.logp_fn = logp_fn
        self.D = self.params[0].shape[0]
        # one step size per chain, a device tensor (captured graphs read it: adapting it is an in-place update)
        self.eps = torch.full((self.D,), float(step_size), dtype=self.params[0].dtype, device=self.params[0].device)
        self.mass = [torch.ones_like(p) if m is None else torch.as_tensor(m, dtype=p.dtype, device=p.device).expand_as(p).clone()
                     for p, m in zip(self.params, mass or [None] * len(self.params))]
        self.generator = generator
        self.n_steps = 0

    # one value + gradient evaluation of all chains
    def _value_and_grad(self, q):
        with torch.enable_grad():
            qs = [x.detach().requires_grad_(True) for x in q]
            lp = self.logp_fn(*qs)
            grads = torch.autograd.grad(lp.sum(), qs)
        return lp.detach(), [g.detach() for g in grads]

    def _kinetic(self, p):
        return sum((0.5 * pi * pi / mi).reshape(self.D, -1).sum(-1) for pi, mi in zip(p, self.mass))

    def _momenta(self):
        return [torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=self.generator) * torch.sqrt(m)
                for q, m in zip(self.params, self.mass)]

    def _rand(self):
        return torch.rand(self.D, dtype=self.eps.dtype, device=self.eps.device, generator=self.generator)

    def _reset_statistics(self):
        self.n_steps = 0

    @torch.no_grad()
    def warmup(self, n_steps, target_accept=0.8, adapt_mass=False, gamma=0.05, t0=10.0, kappa=0.75):
        """``n_steps`` transitions that adapt, per chain, the step size to the acceptance statistic ``target_accept``
        by dual averaging; afterwards the averaged step sizes are kept.  ``adapt_mass``: half way through, the
        diagonal masses become 1 / variance of every parameter element, pooled over chains and the draws so far (the
        chains of a batch target one posterior), and the step-size adaptation restarts.  Nothing here synchronises
        with the host (beyond what ``step`` does); statistics of the sampling phase start after the warm-up."""
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
        self._reset_statistics()
        return self.eps


class HMC(_ChainSampler):
    """``logp_fn(*params) -> (D,)`` log-density of every chain; ``params``: tensors with a leading
    chain dimension D (their values are the chains' starting points and are updated in place).

    ``step()`` runs one trajectory for all chains and returns the (D,) bool accept mask (a device
    tensor: reading it is the caller's choice of when to synchronise).  ``mass``: one tensor per
    parameter (broadcastable to it), default 1.  On a ROCm device the trajectory is replayed as a
    hipGraph (``graph=False`` launches it eagerly); on the CPU it simply runs -- the sampler logic is
    device-agnostic, which is how tests/test_sampling.py checks it without a GPU.
    """

    def __init__(self, logp_fn, params, step_size, n_leapfrog, mass=None, graph=True, generator=None):
        self.L = int(n_leapfrog)
        if self.L < 1 or not float(step_size) > 0:
            raise ValueError("need step_size > 0 and n_leapfrog >= 1")
        self._setup(logp_fn, params, step_size, mass, generator)
        self.n_accept = torch.zeros(self.D, dtype=torch.float64, device=self.params[0].device)
        self._graph = None
        if graph and self.params[0].is_cuda:
            k = len(self.params)
            # static inputs: positions, momenta; static outputs: proposal, its momenta, both log-densities
            q0 = [p.clone() for p in self.params]
            p0 = [torch.zeros_like(p) for p in self.params]
            self._graph = GraphedStep(lambda *a: self._trajectory(list(a[:k]), list(a[k:])), *q0, *p0)

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

    @torch.no_grad()
    def step(self):
        k = len(self.params)
        p0 = self._momenta()
        if self._graph is not None:
            out = self._graph(*self.params, *p0)
        else:
            out = self._trajectory(self.params, p0)
        q1, p1, lp0, lp1 = list(out[:k]), list(out[k:2 * k]), out[2 * k], out[2 * k + 1]
        dH = (lp1 - self._kinetic(p1)) - (lp0 - self._kinetic(p0))       # -(H1 - H0)
        u = self._rand()
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

    def _reset_statistics(self):
        self.n_steps = 0
        self.n_accept.zero_()

    def accept_rate(self):
        """per-chain acceptance fraction so far (device tensor)"""
        return self.n_accept / max(self.n_steps, 1)


def _ckpt_range(n):
    """checkpoint slots that leaf ``n`` (0-based, odd) of a sub-tree closes: the sub-trees ending at n have 2, 4, ...
    leaves, one per trailing 1 bit of n; their first leaves were stored at slots idx_min..idx_max"""
    idx_max = bin(n >> 1).count("1")
    ones, m = 0, n
    while m & 1:
        ones += 1
        m >>= 1
    return idx_max - ones + 1, idx_max


class NUTS(_ChainSampler):
    """No-U-Turn sampler over a batch of chains (multinomial sampling along the trajectory, generalised turning
    criterion, diagonal masses, one step size per chain).

    ``logp_fn(*params) -> (D,)``; ``params``: tensors with a leading chain dimension D (starting points, updated in
    place).  ``step()`` makes one transition of every chain and returns the (D,) tree depths (device tensor).  All
    chains build their trees in lockstep: doubling j adds 2**j leaves to every chain that is still growing, in the
    direction each chain drew for itself; a leaf is ONE batched value + gradient evaluation (on a ROCm device the
    leapfrog step around it is replayed as a hipGraph), chains that have stopped -- turned, diverged -- carry their state
    along unchanged.  The host looks at the device once per doubling ("any chain still growing?").

    After ``step()``: ``last_accept_prob`` (mean over the trajectory's leaves of min(1, exp(H0 - H)): what ``warmup``
    adapts the step sizes to), ``last_depth``, ``last_diverged``, ``last_logp``; ``n_divergent`` counts per chain.
    """

    def __init__(self, logp_fn, params, step_size, max_depth=8, mass=None, graph=True, generator=None,
                 max_energy_error=1000.0):
        self.max_depth = int(max_depth)
        if self.max_depth < 1:
            raise ValueError("need max_depth >= 1")
        self._setup(logp_fn, params, step_size, mass, generator)
        self.max_energy_error = float(max_energy_error)
        dev = self.params[0].device
        self.n_divergent = torch.zeros(self.D, dtype=torch.float64, device=dev)
        self.sum_depth = torch.zeros(self.D, dtype=torch.float64, device=dev)
        self.n_leapfrog = 0          # leaves evaluated so far (per chain slot: masked chains ride along)
        self._lp = self._g = None    # log-density and gradient at the current positions
        # the tree lives on ONE (D, n) array per quantity -- all parameter blocks of a chain side by side -- so that a
        # mask, a dot product or a checkpoint is one launch whatever the number of blocks; logp_fn sees views of it
        self._shapes = [tuple(p.shape) for p in self.params]
        self._sizes = [int(p[0].numel()) for p in self.params]
        self._mflat = self._flat(self.mass)
        self._graph = None
        if graph and self.params[0].is_cuda:
            q = self._flat(self.params)
            self._graph = GraphedStep(self._leaf, q, torch.zeros_like(q), torch.zeros_like(q), self.eps.clone())

    def _flat(self, parts):
        return torch.cat([x.reshape(self.D, -1) for x in parts], dim=1)

    def _parts(self, flat):
        return [x.reshape(shp) for x, shp in zip(torch.split(flat, self._sizes, dim=1), self._shapes)]

    def _value_and_grad_flat(self, q):
        with torch.enable_grad():
            parts = self._parts(q.detach().requires_grad_(True))
            lp = self.logp_fn(*parts)
            grads = torch.autograd.grad(lp.sum(), parts)
        return lp.detach(), self._flat([g.detach() for g in grads])

    def _leaf(self, q, p, g, eps):
        """one leapfrog step of every chain with its own signed step size: (q, p, grad) -> (q', p', grad', logp')"""
        e = eps.unsqueeze(1)
        p = p + 0.5 * e * g
        q = q + e * p / self._mflat
        lp, g = self._value_and_grad_flat(q)
        p = p + 0.5 * e * g
        return q, p, g, lp

    def _turning(self, p_left, p_right, p_sum):
        """generalised U-turn: the trajectory's ends no longer move apart along rho = sum of momenta
        (minus half of each end, Betancourt 2017 / Stan)"""
        rho = (p_sum - 0.5 * (p_left + p_right)) / self._mflat
        return ((p_left * rho).sum(1) <= 0) | ((p_right * rho).sum(1) <= 0)

    @torch.no_grad()
    def step(self):
        self._mflat.copy_(self._flat(self.mass))         # (warm-up may have changed the masses; captured graphs read this)
        q0 = self._flat(self.params)
        if self._lp is None:
            self._lp, self._g = self._value_and_grad_flat(q0)
        lp0, g0 = self._lp, self._g
        kin = lambda p: (0.5 * p * p / self._mflat).sum(1)  # noqa: E731
        w1 = lambda m, a, b: torch.where(m.unsqueeze(1), a, b)  # noqa: E731
        p0 = torch.randn(q0.shape, dtype=q0.dtype, device=q0.device, generator=self.generator) * torch.sqrt(self._mflat)
        H0 = -lp0 + kin(p0)
        valid = torch.isfinite(H0)                         # a chain outside the support (or at a NaN) stays where it is
        H0 = torch.where(valid, H0, torch.zeros_like(H0))
        ql, pl, gl = q0, p0, g0                            # the trajectory's ends
        qr, pr, gr = q0, p0, g0
        p_sum = p0
        log_w = torch.zeros_like(H0)                       # log of the tree's total weight, relative to exp(-H0)
        prop_q, prop_lp, prop_g = q0, lp0, g0
        active = valid
        depth = torch.zeros_like(H0)
        diverged = torch.zeros_like(active)
        acc_sum, acc_n = torch.zeros_like(H0), torch.zeros_like(H0)
        ninf = torch.full_like(H0, float("-inf"))
        zeros = torch.zeros_like(p0)
        for j in range(self.max_depth):
            going_right = self._rand() < 0.5
            eps_s = torch.where(going_right, self.eps, -self.eps)
            qe, pe, ge = w1(going_right, qr, ql), w1(going_right, pr, pl), w1(going_right, gr, gl)
            sub_on = active                                  # still adding leaves to this sub-tree
            sub_turn = torch.zeros_like(active)
            sub_div = torch.zeros_like(active)
            sub_logw = ninf
            sub_psum = zeros
            sub_q, sub_lp, sub_g = qe, lp0, ge
            n_slots = max(j, 1)
            ck_p, ck_sum = [zeros] * n_slots, [zeros] * n_slots
            for n in range(2 ** j):
                out = self._graph(qe, pe, ge, eps_s) if self._graph is not None else self._leaf(qe, pe, ge, eps_s)
                self.n_leapfrog += 1
                qn, pn, gn, lpn = out
                dH = -lpn + kin(pn) - H0
                dH = torch.where(torch.isnan(dH), torch.full_like(dH, float("inf")), dH)
                div = sub_on & (dH > self.max_energy_error)
                ok = sub_on & ~div
                qe, pe, ge = w1(sub_on, qn, qe), w1(sub_on, pn, pe), w1(sub_on, gn, ge)
                acc_sum = acc_sum + torch.where(sub_on, torch.exp(torch.clamp(-dH, max=0.0)), torch.zeros_like(dH))
                acc_n = acc_n + sub_on.to(acc_n.dtype)
                # multinomial sampling within the sub-tree: the new leaf replaces the candidate with probability w / W
                new_logw = torch.logaddexp(sub_logw, -dH)
                take = ok & (torch.log(self._rand()) < (-dH - new_logw))
                sub_q, sub_g = w1(take, qn, sub_q), w1(take, gn, sub_g)
                sub_lp = torch.where(take, lpn, sub_lp)
                sub_logw = torch.where(ok, new_logw, sub_logw)
                sub_psum = w1(ok, sub_psum + pn, sub_psum)
                if n % 2 == 0:
                    slot = bin(n >> 1).count("1")
                    ck_p[slot] = w1(ok, pn, ck_p[slot])
                    ck_sum[slot] = w1(ok, sub_psum, ck_sum[slot])
                else:
                    lo, hi = _ckpt_range(n)
                    for i in range(hi, lo - 1, -1):
                        sub_turn = sub_turn | (ok & self._turning(ck_p[i], pn, sub_psum - ck_sum[i] + ck_p[i]))
                sub_div = sub_div | div
                sub_on = sub_on & ~div & ~sub_turn
            grown = active & ~sub_turn & ~sub_div               # the sub-tree is valid: it joins the tree
            # biased progressive sampling between the old tree and the new half
            take = grown & (torch.log(self._rand()) < (sub_logw - log_w))
            prop_q, prop_g = w1(take, sub_q, prop_q), w1(take, sub_g, prop_g)
            prop_lp = torch.where(take, sub_lp, prop_lp)
            g_r, g_l = grown & going_right, grown & ~going_right
            qr, pr, gr = w1(g_r, qe, qr), w1(g_r, pe, pr), w1(g_r, ge, gr)
            ql, pl, gl = w1(g_l, qe, ql), w1(g_l, pe, pl), w1(g_l, ge, gl)
            p_sum = w1(grown, p_sum + sub_psum, p_sum)
            log_w = torch.where(grown, torch.logaddexp(log_w, sub_logw), log_w)
            depth = depth + active.to(depth.dtype)
            diverged = diverged | (active & sub_div)
            active = grown & ~self._turning(pl, pr, p_sum)
            if j + 1 < self.max_depth and not bool(active.any()):   # (the one host synchronisation per doubling)
                break
        for x, y in zip(self.params, self._parts(prop_q)):
            x.copy_(y)
        self._lp, self._g = prop_lp, prop_g.clone()
        self.n_steps += 1
        self.last_logp = prop_lp
        self.last_depth = depth
        self.last_diverged = diverged
        self.last_accept_prob = acc_sum / torch.clamp(acc_n, min=1.0)
        self.n_divergent += diverged.to(self.n_divergent.dtype)
        self.sum_depth += depth
        return depth

    def _reset_statistics(self):
        self.n_steps = 0
        self.n_divergent.zero_()
        self.sum_depth.zero_()
        self.n_leapfrog = 0

    def mean_depth(self):
        """per-chain mean tree depth so far (device tensor)"""
        return self.sum_depth / max(self.n_steps, 1)
