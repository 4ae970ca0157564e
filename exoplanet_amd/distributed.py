"""Multi-GPU: posterior draws / chains sharded over ranks, one process per GPU.

The reference has no parallelism beyond PyMC's one-process-per-chain
(/root/reference/docs/user/multiprocessing.rst:6-8).  Here the D draws of a
batch are independent units: rank g of W evaluates draws [lo_g, hi_g) on its own
GPU (t, y, diag are replicated, a few MB), parameter gradients stay on the
owning rank, and the ONLY exchange is the per-draw log-likelihood vector -- one
collective of D doubles (<= 8 KiB at D = 1024) over RCCL / xGMI (torch backend
"nccl"); latency-bound, so it is issued exactly once per evaluation.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard", "LoglikeExchange", "gather_loglike", "sharded_log_likelihood"]


def _rank_world(group=None):
    if not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_bounds(n_draw, rank=None, world=None, group=None):
    """[lo, hi) of this rank's contiguous block of draws (blocks differ by <= 1).  ``rank`` /
    ``world`` default to this process's position in ``group`` (the default group if None)."""
    r, w = _rank_world(group)
    if world is None:
        world = w
    if rank is None:
        rank = r
    base, extra = divmod(n_draw, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(x, n_draw=None, rank=None, world=None, group=None):
    """slice the leading (draw) dimension of a tensor / pytree of tensors for this rank"""
    if isinstance(x, dict):
        return {k: shard(v, n_draw, rank, world, group) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(shard(v, n_draw, rank, world, group) for v in x)
    n = x.shape[0] if n_draw is None else n_draw
    lo, hi = shard_bounds(n, rank, world, group)
    return x[lo:hi]


class LoglikeExchange:
    """The one collective of a step: every rank receives the full (n_draw,) vector of per-draw
    scalars.  The output buffer is allocated once; a call issues exactly ONE collective and no
    other device work when the shards are equal (``all_gather_into_tensor`` straight from the
    caller's tensor -- e.g. the static output of a replayed hipGraph); ragged shards fall back
    to "own slice of a zero vector + all-reduce(SUM)".  ``start`` / ``finish`` are the pipelined
    form of the same exchange: the collective of a step overlaps the kernels of the next one, and every ``start``
    hands back the previous step's completed vector.  Rank and world size are those of
    ``group``.  A single process (or an uninitialised process group) just copies."""

    def __init__(self, n_draw, device, dtype=torch.float64, group=None, force_collective=False):
        """``force_collective``: issue the collective even in a group of ONE rank (instead of a copy) -- how a 1-GPU box
        exercises and traces the RCCL path (bench.py, EXO_BENCH_FORCE_DIST=1)"""
        self.group = group
        self.rank, self.world = _rank_world(group)
        self._force = bool(force_collective) and dist.is_initialized()
        self.n_draw = int(n_draw)
        self.lo, self.hi = shard_bounds(self.n_draw, self.rank, self.world)
        self.equal = self.n_draw % self.world == 0
        self.device, self.dtype = device, dtype
        self.out = torch.empty(self.n_draw, dtype=dtype, device=device)
        # pipelined form (start / finish): two private copies of the rank's slice, two outputs, the collectives in
        # flight -- allocated by the first start(), a synchronous exchange never pays for them
        self._stage = self._outs = None
        self._work = [None, None]
        self._last = None

    def _wait(self, k):
        if self._work[k] is not None:
            self._work[k].wait()       # (RCCL: the current stream waits for the collective; no host block)
            self._work[k] = None

    def start(self, local):
        """The same collective, PIPELINED: it is issued asynchronously from a private copy of ``local`` and may still be
        in flight while the next step's kernels run -- nothing in a step depends on the other ranks' scalars, a sampler
        reads them for adaptation and logging.  ``local`` may be overwritten as soon as this returns (e.g. the static
        output of a replayed hipGraph).  RETURNS the completed full vector of the PREVIOUS ``start`` (None on the first
        call): the consumer of step k's scalars runs one step behind, which is what lets the collective overlap.  That
        vector is valid until the next ``start`` returns -- use it (or copy it) before then; two buffers alternate.
        ``finish()`` waits for the collective still in flight and returns the vector of the LATEST step."""
        local = local.detach()
        if local.shape != (self.hi - self.lo,):
            raise ValueError(f"rank owns draws [{self.lo},{self.hi}) but got a tensor of shape {tuple(local.shape)}")
        if self._stage is None:
            self._stage = [torch.empty(self.hi - self.lo, dtype=self.dtype, device=self.device) for _ in range(2)]
            self._outs = [torch.empty(self.n_draw, dtype=self.dtype, device=self.device) for _ in range(2)]
        prev = self._last
        prev_out = None
        if prev is not None:
            self._wait(prev)
            prev_out = self._outs[prev]
        k = 0 if prev is None else 1 - prev
        self._wait(k)                  # (issued two steps ago and already waited for when it was handed out: a no-op)
        stage, out = self._stage[k], self._outs[k]
        stage.copy_(local)
        if self.world == 1 and not self._force:
            out.copy_(stage)
        elif self.equal:
            self._work[k] = dist.all_gather_into_tensor(out, stage, group=self.group, async_op=True)
        else:
            out.zero_()
            out[self.lo:self.hi] = stage
            self._work[k] = dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._last = k
        return prev_out

    def finish(self):
        """wait for the collectives in flight; the full vector of the most recent ``start`` (None if there was none)"""
        for k in (0, 1):
            self._wait(k)
        return None if self._last is None else self._outs[self._last]

    def __call__(self, local):
        local = local.detach()
        if local.shape != (self.hi - self.lo,):
            raise ValueError(f"rank owns draws [{self.lo},{self.hi}) but got a tensor of shape {tuple(local.shape)}")
        if self.world == 1 and not self._force:
            self.out.copy_(local)
        elif self.equal:
            dist.all_gather_into_tensor(self.out, local.contiguous(), group=self.group)
        else:
            self.out.zero_()
            self.out[self.lo:self.hi] = local
            dist.all_reduce(self.out, op=dist.ReduceOp.SUM, group=self.group)
        return self.out


def gather_loglike(local, n_draw, group=None):
    """every rank receives the full (n_draw,) vector (one collective; see LoglikeExchange).
    A fresh output per call: loops should keep a LoglikeExchange instead."""
    return LoglikeExchange(n_draw, local.device, local.dtype, group)(local)


def sharded_log_likelihood(loglike_fn, params, n_draw, group=None):
    """Evaluate ``loglike_fn(params_shard) -> (n_local,)`` on this rank's draws and
    return (full log-likelihood vector on every rank, local log-likelihoods with
    their autograd graph).  Gradients never leave the owning rank."""
    local_params = shard(params, n_draw, group=group)
    local = loglike_fn(local_params)
    return gather_loglike(local, n_draw, group), local
