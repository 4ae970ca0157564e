"""Kernel terms: each exposes ``get_coefficients()`` -> (ar, cr, ac, bc, cc, dc),
the representation celerite2's ``Term.get_coefficients`` uses:

    k(tau) = sum_r ar e^{-cr tau} + sum_c e^{-cc tau} (ac cos(dc tau) + bc sin(dc tau))

Parameters may carry leading draw dimensions; coefficients come back with
shape ``(*draws, n_terms)``.  Everything is torch, so autograd carries the
kernel hyper-parameters.
"""
import math

import torch

from .. import ops
from ..orbits.keplerian import as_tensor

__all__ = ["Term", "TermSum", "RealTerm", "ComplexTerm", "SHOTerm", "RotationTerm", "Matern32Term"]


def _empty(like):
    return torch.zeros(like.shape[:-1] + (0,), dtype=torch.float64, device=like.device)


class Term:
    def get_coefficients(self):
        raise NotImplementedError

    def __add__(self, other):
        return TermSum(self, other)

    def __radd__(self, other):
        return TermSum(other, self)

    def pair_coefficients(self):
        """(ar, cr, pairs, kind) -- what the kernels take (include/exoplanet_amd.h): real terms
        ``ar, cr`` (*draws, Jr); pair slots ``pairs`` (*draws, Jc, 4), each a complex term
        (a, b, c, d) or, where ``kind`` (*draws, Jc; int32, or None = all complex) is 1, two real
        terms (a1, c1, a2, c2).  Never synchronises with the host: usable inside a captured step."""
        ar, cr, ac, bc, cc, dc = self.get_coefficients()
        return ar, cr, torch.stack(torch.broadcast_tensors(ac, bc, cc, dc), dim=-1), None

    def get_value(self, tau):
        """k(tau), dense; for tests and plotting"""
        ar, cr, ac, bc, cc, dc = self.get_coefficients()
        tau = as_tensor(tau, ar).abs().unsqueeze(-1)
        k = (ar * torch.exp(-cr * tau)).sum(-1)
        k = k + (torch.exp(-cc * tau) * (ac * torch.cos(dc * tau) + bc * torch.sin(dc * tau))).sum(-1)
        return k


class TermSum(Term):
    def __init__(self, *terms):
        flat = []
        for t in terms:
            flat.extend(t.terms if isinstance(t, TermSum) else [t])
        if any(not isinstance(t, Term) for t in flat):
            raise TypeError("terms must be exoplanet_amd.gp.terms.Term instances")
        self.terms = tuple(flat)

    def _fused_sho(self):
        """a sum of SHO terms on the device: every term's pair slot from ONE launch (ops.sho_coefficients_multi), written where
        the kernels read it -- per term and step this was two launches, a slice of a concatenation and a strided copy of the
        cotangent; None when the sum is anything else"""
        if not (1 < len(self.terms) <= ops.SHO_MAX_TERMS) or not all(type(t) is SHOTerm for t in self.terms):
            return None
        raws = [t._raw for t in self.terms]
        if not all(x.is_cuda for amp, freq, damp, _ in raws for x in (amp, freq, damp)) or len({t.eps for t in self.terms}) != 1:
            return None
        shape = torch.broadcast_shapes(*[x.shape for amp, freq, damp, _ in raws for x in (amp, freq, damp)])
        flat = [(amp.expand(shape).reshape(-1), freq.expand(shape).reshape(-1), damp.expand(shape).reshape(-1), fl)
                for amp, freq, damp, fl in raws]
        coef, kind = ops.sho_coefficients_multi(flat, self.terms[0].eps)
        e = torch.zeros(shape + (0,), dtype=torch.float64, device=coef.device)
        T = len(self.terms)
        return e, e, coef.reshape(shape + (T, 4)), kind.reshape(shape + (T,))

    def get_coefficients(self):
        parts = [t.get_coefficients() for t in self.terms]
        batch = torch.broadcast_shapes(*[p[0].shape[:-1] for p in parts])
        out = []
        for i in range(6):
            out.append(torch.cat([p[i].expand(batch + (p[i].shape[-1],)) for p in parts], dim=-1))
        return tuple(out)

    def pair_coefficients(self):
        fused = self._fused_sho()
        if fused is not None:
            return fused
        parts = [t.pair_coefficients() for t in self.terms]
        batch = torch.broadcast_shapes(*[p[0].shape[:-1] for p in parts], *[p[2].shape[:-2] for p in parts])
        ar = torch.cat([p[0].expand(batch + (p[0].shape[-1],)) for p in parts], dim=-1)
        cr = torch.cat([p[1].expand(batch + (p[1].shape[-1],)) for p in parts], dim=-1)
        pairs = torch.cat([p[2].expand(batch + tuple(p[2].shape[-2:])) for p in parts], dim=-2)
        if all(p[3] is None for p in parts):
            return ar, cr, pairs, None
        kinds = [torch.zeros(batch + (p[2].shape[-2],), dtype=torch.int32, device=pairs.device) if p[3] is None
                 else p[3].expand(batch + (p[3].shape[-1],)) for p in parts]
        return ar, cr, pairs, torch.cat(kinds, dim=-1)


class RealTerm(Term):
    """k = a exp(-c tau)"""

    def __init__(self, *, a, c):
        self.a = as_tensor(a)
        self.c = as_tensor(c, self.a)

    def get_coefficients(self):
        a, c = torch.broadcast_tensors(self.a, self.c)
        a, c = a.unsqueeze(-1), c.unsqueeze(-1)
        e = _empty(a)
        return a, c, e, e, e, e


class ComplexTerm(Term):
    """k = exp(-c tau) (a cos(d tau) + b sin(d tau))"""

    def __init__(self, *, a, b, c, d):
        self.a = as_tensor(a)
        self.b, self.c, self.d = as_tensor(b, self.a), as_tensor(c, self.a), as_tensor(d, self.a)

    def get_coefficients(self):
        a, b, c, d = [x.unsqueeze(-1) for x in torch.broadcast_tensors(self.a, self.b, self.c, self.d)]
        e = _empty(a)
        return e, e, a, b, c, d


class SHOTerm(Term):
    """Stochastically driven damped harmonic oscillator.

    Parameterised like celerite2's SHOTerm: amplitude ``S0`` or ``sigma``;
    frequency ``w0`` or ``rho`` (undamped period); damping ``Q`` or ``tau``.
    Q < 1/2 gives two real terms, Q >= 1/2 one complex term.  Either way the term occupies one
    "pair slot" (two state indices) of the kernels, with the kind decided per draw ON THE DEVICE
    (``pair_coefficients``): a batch of draws may straddle Q = 1/2, and nothing here synchronises
    with the host -- a whole log-likelihood step can be captured in a hipGraph.  ``get_coefficients``
    (celerite2's six arrays, whose shapes depend on the regime) needs all draws on one side and
    looks at Q on the host: it is for plots and tests, not for the sampling loop.
    """

    def __init__(self, *, S0=None, sigma=None, w0=None, rho=None, Q=None, tau=None, eps=1e-5):
        if (w0 is None) == (rho is None):
            raise ValueError("exactly one of w0 and rho must be given")
        if (Q is None) == (tau is None):
            raise ValueError("exactly one of Q and tau must be given")
        if (S0 is None) == (sigma is None):
            raise ValueError("exactly one of S0 and sigma must be given")
        self.eps = eps
        # the parameters as given (the fused coefficient op takes any parameterisation) ...
        freq = as_tensor(w0 if w0 is not None else rho)
        self._raw = (as_tensor(S0 if S0 is not None else sigma, freq), freq, as_tensor(Q if Q is not None else tau, freq),
                     (ops.SHO_SIGMA if S0 is None else 0) | (ops.SHO_RHO if w0 is None else 0) | (ops.SHO_TAU if Q is None else 0))
        self._s0wq = None

    def _derived(self):
        """(S0, w0, Q), celerite2's attribute names, computed when first asked for (the fused path never does)"""
        if self._s0wq is None:
            amp, freq, damp, flags = self._raw
            w0 = 2 * math.pi / freq if flags & ops.SHO_RHO else freq
            Q = 0.5 * w0 * damp if flags & ops.SHO_TAU else damp
            S0 = amp ** 2 / (w0 * Q) if flags & ops.SHO_SIGMA else amp
            self._s0wq = (S0, w0, Q)
        return self._s0wq

    S0 = property(lambda self: self._derived()[0])
    w0 = property(lambda self: self._derived()[1])
    Q = property(lambda self: self._derived()[2])

    def pair_coefficients(self):
        amp, freq, damp, flags = self._raw
        if amp.is_cuda and freq.is_cuda and damp.is_cuda:
            # one launch (and one for the reverse) instead of ~35 elementwise kernels per term and step
            amp, freq, damp = torch.broadcast_tensors(amp, freq, damp)
            shape = amp.shape
            coef, kind = ops.sho_coefficients(amp.reshape(-1), freq.reshape(-1), damp.reshape(-1), flags, self.eps)
            e = _empty(amp.unsqueeze(-1))
            return e, e, coef.reshape(shape + (1, 4)), kind.reshape(shape + (1,))
        S0, w0, Q = torch.broadcast_tensors(self.S0, self.w0, self.Q)
        over = Q.detach() < 0.5
        # both parameterisations, each clamped inside its own domain, the draw's regime selects
        fo = torch.sqrt(torch.clamp(1.0 - 4.0 * Q ** 2, min=self.eps))
        fu = torch.sqrt(torch.clamp(4.0 * Q ** 2 - 1.0, min=self.eps))
        a = S0 * w0 * Q
        c = 0.5 * w0 / Q
        two_real = torch.stack([0.5 * a * (1.0 + 1.0 / fo), c * (1.0 - fo), 0.5 * a * (1.0 - 1.0 / fo), c * (1.0 + fo)], dim=-1)
        one_complex = torch.stack([a, a / fu, c, c * fu], dim=-1)
        pairs = torch.where(over.unsqueeze(-1), two_real, one_complex).unsqueeze(-2)
        e = _empty(S0.unsqueeze(-1))
        return e, e, pairs, over.to(torch.int32).unsqueeze(-1)

    def get_coefficients(self):
        S0, w0, Q = torch.broadcast_tensors(self.S0, self.w0, self.Q)
        over = Q.detach() < 0.5
        if bool(over.any()) and not bool(over.all()):
            raise ValueError("SHOTerm: a batch mixes Q < 1/2 and Q >= 1/2 draws (different celerite state layouts)")
        S0, w0, Q = S0.unsqueeze(-1), w0.unsqueeze(-1), Q.unsqueeze(-1)
        e = _empty(S0)
        if bool(over.all()) and over.numel() > 0:
            f = torch.sqrt(torch.clamp(1.0 - 4.0 * Q ** 2, min=self.eps))
            pm = torch.tensor([1.0, -1.0], dtype=torch.float64, device=S0.device)
            ar = 0.5 * S0 * w0 * Q * (1.0 + pm / f)
            cr = 0.5 * w0 / Q * (1.0 - pm * f)
            return ar, cr, e, e, e, e
        f = torch.sqrt(torch.clamp(4.0 * Q ** 2 - 1.0, min=self.eps))
        a = S0 * w0 * Q
        c = 0.5 * w0 / Q
        return e, e, a, a / f, c, c * f


class RotationTerm(TermSum):
    """Two SHOs at the rotation period and its first harmonic (celerite2's RotationTerm)."""

    def __init__(self, *, sigma, period, Q0, dQ, f):
        sigma, period = as_tensor(sigma), as_tensor(period)
        Q0, dQ, f = as_tensor(Q0, sigma), as_tensor(dQ, sigma), as_tensor(f, sigma)
        amp = sigma ** 2 / (1 + f)
        Q1 = 0.5 + Q0 + dQ
        w1 = 4 * math.pi * Q1 / (period * torch.sqrt(4 * Q1 ** 2 - 1))
        S1 = amp / (w1 * Q1)
        Q2 = 0.5 + Q0
        w2 = 8 * math.pi * Q2 / (period * torch.sqrt(4 * Q2 ** 2 - 1))
        S2 = f * amp / (w2 * Q2)
        super().__init__(SHOTerm(S0=S1, w0=w1, Q=Q1), SHOTerm(S0=S2, w0=w2, Q=Q2))


class Matern32Term(Term):
    """celerite approximation of the Matern-3/2 kernel (one complex term, small eps)"""

    def __init__(self, *, sigma, rho, eps=0.01):
        self.sigma = as_tensor(sigma)
        self.rho = as_tensor(rho, self.sigma)
        self.eps = eps

    def get_coefficients(self):
        sigma, rho = torch.broadcast_tensors(self.sigma, self.rho)
        w0 = (math.sqrt(3.0) / rho).unsqueeze(-1)
        S0 = (sigma ** 2).unsqueeze(-1) / w0
        e = _empty(w0)
        return e, e, w0 * S0, w0 * w0 * S0 / self.eps, w0, self.eps + torch.zeros_like(w0)
