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
        self._shapes = [tuple(p.shape) for p in self.params]
        self._sizes = [int(p[0].numel()) for p in self.params]

    # one value + gradient evaluation of all chains
    def _value_and_grad(self, q):
        with torch.enable_grad():
            qs = [x.detach().requires_grad_(True) for x in q]
            lp = self.logp_fn(*qs)
            grads = torch.autograd.grad(lp.sum(), qs)
        return lp.detach(), [g.detach() for g in grads]

    def _kinetic(self, p):
        return sum((0.5 * pi * pi / mi).reshape(self.D, -1).sum(-1) for pi, mi in zip(p, self.mass))

    # all parameter blocks of a chain side by side in ONE (D, n) array: an update of positions or momenta is one launch
    # whatever the number of blocks; logp_fn sees views of it
    def _flat(self, parts, out=None):
        return torch.cat([x.reshape(self.D, -1) for x in parts], dim=1, out=out)

    def _parts(self, flat):
        return [x.reshape(shp) for x, shp in zip(torch.split(flat, self._sizes, dim=1), self._shapes)]

    def _value_and_grad_flat(self, q, grad_out=None):
        """(logp (D,), its gradient as a flat (D, n) array); ``grad_out``: write the gradient there (no copy afterwards)"""
        with torch.enable_grad():
            parts = self._parts(q.detach().requires_grad_(True))
            lp = self.logp_fn(*parts)
            # d(sum of the chains' log-densities): a unit cotangent kept from call to call (no reduction kernel, no fill)
            ones = getattr(self, "_unit", None)
            if ones is None or ones.shape != lp.shape or ones.device != lp.device:
                ones = self._unit = torch.ones_like(lp).detach()
            grads = torch.autograd.grad(lp, parts, grad_outputs=ones)
        return lp.detach(), self._flat([g.detach() for g in grads], out=grad_out)

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
        log_eps0 = torch.log(self.eps)
        for it in range(n_steps):
            self.step()
            m += 1
            # a chain that made no move this step for a reason that is not its step size (it sits outside the support:
            # no valid leaf) reports the target itself -- its step size is left alone instead of being driven to zero
            acc = torch.where(self.last_adapt_ok, self.last_accept_prob, torch.full_like(self.last_accept_prob, target_accept))
            Hbar = (1.0 - 1.0 / (m + t0)) * Hbar + (target_accept - acc) / (m + t0)
            log_eps = torch.clamp(mu - (m ** 0.5 / gamma) * Hbar, min=log_eps0 - 30.0, max=log_eps0 + 30.0)
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
        self._mflat = self._flat(self.mass)
        self._graph = None
        if graph and self.params[0].is_cuda:
            # static inputs: positions, momenta (flat); static outputs: proposal, its momenta, both log-densities
            q0 = self._flat(self.params)
            self._graph = GraphedStep(self._trajectory, q0, torch.zeros_like(q0))

    def _trajectory(self, q, p):
        """leapfrog on the flat (D, n) arrays: (q, p) -> (q', p', logp(q), logp(q'))"""
        lp0, g = self._value_and_grad_flat(q)
        lp = lp0
        e = self.eps.unsqueeze(1)
        eh, em = 0.5 * e, e / self._mflat
        # (the closing half step of one leapfrog step and the opening half step of the next use the same gradient: one
        # fused multiply-add each for p and q per step -- every elementwise kernel here is a launch between two likelihoods)
        p = torch.addcmul(p, eh, g)
        for i in range(self.L):
            q = torch.addcmul(q, em, p)
            lp, g = self._value_and_grad_flat(q)
            p = torch.addcmul(p, e if i + 1 < self.L else eh, g)
        return q, p, lp0, lp

    @torch.no_grad()
    def step(self):
        self._mflat.copy_(self._flat(self.mass))          # (warm-up may have changed the masses; the captured graph reads this)
        q0 = self._flat(self.params)
        p0 = torch.randn(q0.shape, dtype=q0.dtype, device=q0.device, generator=self.generator) * torch.sqrt(self._mflat)
        q1, p1, lp0, lp1 = self._graph(q0, p0) if self._graph is not None else self._trajectory(q0, p0)
        kin = lambda p: (0.5 * p * p / self._mflat).sum(1)  # noqa: E731
        dH = (lp1 - kin(p1)) - (lp0 - kin(p0))                           # -(H1 - H0)
        u = self._rand()
        accept = torch.log(u) < dH                                        # NaN / -inf proposals are rejected
        qnew = torch.where(accept.unsqueeze(1), q1, q0)
        for x, y in zip(self.params, self._parts(qnew)):
            x.copy_(y)
        self.n_steps += 1
        self.n_accept += accept.to(self.n_accept.dtype)
        self.last_logp = torch.where(accept, lp1, lp0)
        # acceptance probability min(1, exp(-dH)) of every chain (NaN proposals: 0): what step-size adaptation feeds on
        self.last_accept_prob = torch.nan_to_num(torch.exp(torch.clamp(dH, max=0.0)), nan=0.0)
        self.last_adapt_ok = torch.isfinite(lp0)           # (a chain outside the support says nothing about its step size)
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
        self._mflat = self._flat(self.mass)
        S = self._S = max(self.max_depth - 1, 1)
        q = self._flat(self.params)
        zq, zd = torch.zeros_like(q), torch.zeros(self.D, dtype=q.dtype, device=dev)
        zb = torch.zeros(self.D, dtype=torch.bool, device=dev)
        n_rows = 2 + 2 ** (self.max_depth - 1)
        # static buffers: the sub-tree being built, the trajectory, the doubling's random numbers (row 0 directions, row 1
        # the merge, row 2 + k leaf k), the index of the current leaf
        self._st = dict(qe=q.clone(), pe=zq.clone(), ge=zq.clone(), eps=self.eps.clone(), on=zb.clone(), H0=zd.clone(),
                        logw=zd.clone(), psum=zq.clone(), sq=zq.clone(), sg=zq.clone(), slp=zd.clone(), turn=zb.clone(),
                        div=zb.clone(), acc=zd.clone(), accn=zd.clone(),
                        ckp=torch.zeros((S,) + tuple(q.shape), dtype=q.dtype, device=dev),
                        cks=torch.zeros((S,) + tuple(q.shape), dtype=q.dtype, device=dev),
                        qn=zq.clone(), ph=zq.clone(), gn=zq.clone(), lpn=zd.clone(),
                        ql=zq.clone(), pl=zq.clone(), gl=zq.clone(), qr=zq.clone(), pr=zq.clone(), gr=zq.clone(), tsum=zq.clone(),
                        logW=zd.clone(), propq=zq.clone(), propg=zq.clone(), proplp=zd.clone(), active=zb.clone(),
                        diverged=zb.clone(), going=zb.clone(), depth=zd.clone(),
                        R=torch.full((n_rows, self.D), 0.5, dtype=q.dtype, device=dev),
                        leaf=torch.zeros(1, dtype=torch.int32, device=dev))
        self._native = None
        if self.params[0].is_cuda:
            # on the device the tree is built by four kernels around the likelihood (exo_nuts_f64)
            import ctypes

            st = self._st
            order = ("qe", "pe", "ge", "eps", "on", "H0", "logw", "psum", "sq", "sg", "slp", "turn", "div", "acc", "accn", "ckp",
                     "cks", None, "qn", "ph", "gn", "lpn", "ql", "pl", "gl", "qr", "pr", "gr", "tsum", "logW", "propq", "propg",
                     "proplp", "active", "diverged", "going", "depth", "eps_abs", "R", "leaf")
            ptrs = [self._mflat.data_ptr() if k is None else (self.eps.data_ptr() if k == "eps_abs" else st[k].data_ptr())
                    for k in order]
            self._native = ((ctypes.c_void_p * len(ptrs))(*ptrs), int(q.shape[1]), S)
        self._graph = None
        if graph and self.params[0].is_cuda:
            self._graph = GraphedStep(lambda *a: self._leaf_update(), *[v for v in self._st.values()])

    def _phase(self, phase):
        from . import _lib

        ptrs, n, S = self._native
        dev = self._st["qe"].device
        with torch.cuda.device(dev):
            _lib.check(_lib.load().exo_nuts_f64(ptrs, self.D, n, S, self.max_energy_error, phase,
                                                torch.cuda.current_stream(dev).cuda_stream), "exo_nuts_f64")

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

    # ---- the three steps of a doubling, on the static buffers: native kernels on the device, torch otherwise --------
    def _begin_doubling(self):
        st = self._st
        if self._native is not None:
            return self._phase(2)
        w1 = lambda m, a, b: torch.where(m.unsqueeze(1), a, b)  # noqa: E731
        right = st["R"][0] < 0.5
        st["going"].copy_(right)
        st["eps"].copy_(torch.where(right, self.eps, -self.eps))
        qe, pe, ge = w1(right, st["qr"], st["ql"]), w1(right, st["pr"], st["pl"]), w1(right, st["gr"], st["gl"])
        for k, v in (("qe", qe), ("pe", pe), ("ge", ge), ("sq", qe), ("sg", ge), ("slp", st["proplp"]), ("on", st["active"])):
            st[k].copy_(v)
        st["logw"].fill_(float("-inf"))
        st["psum"].zero_(); st["turn"].zero_(); st["div"].zero_()
        st["leaf"].fill_(-1)

    def _leaf_update(self):
        """one leaf of the sub-tree, in place: leapfrog step of the moving end, energy error, divergence, multinomial
        candidate, momentum sums, checkpoint write (even leaves) or turning checks (odd leaves), the chains that go on.
        Eager, or captured once and replayed per leaf (the leaf's index is counted on the device)."""
        st = self._st
        if self._native is not None:
            self._phase(0)
            lp, _ = self._value_and_grad_flat(st["qn"], grad_out=st["gn"])     # (the gradient lands in the leaf's buffer)
            st["lpn"].copy_(lp)
            self._phase(1)
            return st["on"]
        st["leaf"].add_(1)
        n = int(st["leaf"])                       # (the torch statement runs on the CPU: reading the index costs nothing)
        qn, pn, gn, lpn = self._leaf(st["qe"], st["pe"], st["ge"], st["eps"])
        with torch.no_grad():
            on = st["on"]
            dH = -lpn + (0.5 * pn * pn / self._mflat).sum(1) - st["H0"]
            dH = torch.where(torch.isnan(dH), torch.full_like(dH, float("inf")), dH)
            div = on & (dH > self.max_energy_error)
            ok = on & ~div
            m1 = on.unsqueeze(1)
            qe, pe, ge = torch.where(m1, qn, st["qe"]), torch.where(m1, pn, st["pe"]), torch.where(m1, gn, st["ge"])
            acc = st["acc"] + torch.where(on, torch.exp(torch.clamp(-dH, max=0.0)), torch.zeros_like(dH))
            accn = st["accn"] + on.to(dH.dtype)
            # multinomial sampling within the sub-tree: the new leaf replaces the candidate with probability w / W
            new_logw = torch.logaddexp(st["logw"], -dH)
            take = ok & (torch.log(st["R"][2 + n]) < (-dH - new_logw))
            t1 = take.unsqueeze(1)
            sq, sg = torch.where(t1, qn, st["sq"]), torch.where(t1, gn, st["sg"])
            slp = torch.where(take, lpn, st["slp"])
            logw = torch.where(ok, new_logw, st["logw"])
            psum = torch.where(ok.unsqueeze(1), st["psum"] + pn, st["psum"])
            turn = st["turn"]
            if n % 2 == 0:                        # its momentum and the sum so far: checkpoint of the sub-sub-trees that start here
                slot = bin(n >> 1).count("1")
                wm = ok.unsqueeze(1)
                st["ckp"][slot].copy_(torch.where(wm, pn, st["ckp"][slot]))
                st["cks"][slot].copy_(torch.where(wm, psum, st["cks"][slot]))
            else:                                 # it closes the sub-sub-trees of 2, 4, ... leaves that end here
                lo, hi = _ckpt_range(n)
                for i in range(hi, lo - 1, -1):
                    inner = psum - st["cks"][i] + st["ckp"][i]
                    turn = turn | (ok & self._turning(st["ckp"][i], pn, inner))
            sdiv = st["div"] | div
            on2 = on & ~div & ~turn
            for k, v in (("qe", qe), ("pe", pe), ("ge", ge), ("acc", acc), ("accn", accn), ("sq", sq), ("sg", sg), ("slp", slp),
                         ("logw", logw), ("psum", psum), ("turn", turn), ("div", sdiv), ("on", on2)):
                st[k].copy_(v)
        return st["on"]

    def _merge(self):
        st = self._st
        if self._native is not None:
            return self._phase(3)
        w1 = lambda m, a, b: torch.where(m.unsqueeze(1), a, b)  # noqa: E731
        active, right = st["active"].clone(), st["going"]
        grown = active & ~st["turn"] & ~st["div"]               # the sub-tree is valid: it joins the trajectory
        # biased progressive sampling between the old trajectory and the new half
        take = grown & (torch.log(st["R"][1]) < (st["logw"] - st["logW"]))
        st["propq"].copy_(w1(take, st["sq"], st["propq"])); st["propg"].copy_(w1(take, st["sg"], st["propg"]))
        st["proplp"].copy_(torch.where(take, st["slp"], st["proplp"]))
        g_r, g_l = grown & right, grown & ~right
        for end, m in (("r", g_r), ("l", g_l)):
            for a, b in (("q", "qe"), ("p", "pe"), ("g", "ge")):
                st[a + end].copy_(w1(m, st[b], st[a + end]))
        st["tsum"].copy_(w1(grown, st["tsum"] + st["psum"], st["tsum"]))
        st["logW"].copy_(torch.where(grown, torch.logaddexp(st["logW"], st["logw"]), st["logW"]))
        st["depth"].add_(active.to(st["depth"].dtype))
        st["diverged"].copy_(st["diverged"] | (active & st["div"]))
        st["active"].copy_(grown & ~self._turning(st["pl"], st["pr"], st["tsum"]))

    @torch.no_grad()
    def step(self):
        st = self._st
        self._mflat.copy_(self._flat(self.mass))         # (warm-up may have changed the masses; captured graphs read this)
        q0 = self._flat(self.params)
        if self._lp is None:
            self._lp, self._g = self._value_and_grad_flat(q0)
        lp0, g0 = self._lp, self._g
        p0 = torch.randn(q0.shape, dtype=q0.dtype, device=q0.device, generator=self.generator) * torch.sqrt(self._mflat)
        H0 = -lp0 + (0.5 * p0 * p0 / self._mflat).sum(1)
        valid = torch.isfinite(H0)                         # a chain outside the support (or at a NaN) stays where it is
        for k, v in (("ql", q0), ("qr", q0), ("pl", p0), ("pr", p0), ("gl", g0), ("gr", g0), ("tsum", p0), ("propq", q0),
                     ("propg", g0), ("proplp", lp0), ("active", valid), ("H0", torch.where(valid, H0, torch.zeros_like(H0)))):
            st[k].copy_(v)
        for k in ("logW", "depth", "diverged", "acc", "accn"):   # (logW: log of the trajectory's weight relative to exp(-H0))
            st[k].zero_()
        for j in range(self.max_depth):
            # the doubling's random numbers: directions, the merge, one row per leaf
            torch.rand(2 + 2 ** j, self.D, dtype=q0.dtype, device=q0.device, generator=self.generator, out=st["R"][:2 + 2 ** j])
            self._begin_doubling()
            for _ in range(2 ** j):
                if self._graph is not None:
                    self._graph()
                else:
                    self._leaf_update()
                self.n_leapfrog += 1
            self._merge()
            if j + 1 < self.max_depth and not bool(st["active"].any()):   # (the one host synchronisation per doubling)
                break
        for x, y in zip(self.params, self._parts(st["propq"])):
            x.copy_(y)
        self._lp, self._g = st["proplp"].clone(), st["propg"].clone()
        self.n_steps += 1
        self.last_logp = self._lp
        self.last_depth = st["depth"].clone()
        self.last_diverged = st["diverged"].clone()
        self.last_accept_prob = st["acc"] / torch.clamp(st["accn"], min=1.0)
        self.last_adapt_ok = st["accn"] > 0                # (no leaf at all: the chain started outside the support)
        self.n_divergent += self.last_diverged.to(self.n_divergent.dtype)
        self.sum_depth += self.last_depth
        return self.last_depth

    def _reset_statistics(self):
        self.n_steps = 0
        self.n_divergent.zero_()
        self.sum_depth.zero_()
        self.n_leapfrog = 0

    def mean_depth(self):
        """per-chain mean tree depth so far (device tensor)"""
        return self.sum_depth / max(self.n_steps, 1)
