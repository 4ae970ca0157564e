"""Multi-GPU: posterior draws / chains sharded over ranks, one process per GPU.

The reference has no parallelism beyond PyMC's one-process-per-chain
(/root/reference/docs/user/multiprocessing.rst:6-8).  Here the D draws of a
batch are independent units: rank g of W evaluates draws [lo_g, hi_g) on its own
GPU (t, y, diag are replicated, a few MB), parameter gradients stay on the
owning rank, and the ONLY exchange is the per-draw log-likelihood vector -- one
all-reduce of D doubles (<= 8 KiB at D = 1024) over RCCL / xGMI (torch backend
"nccl"); latency-bound, so it is issued exactly once per evaluation.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard", "gather_loglike", "sharded_log_likelihood"]


def shard_bounds(n_draw, rank=None, world=None):
    """[lo, hi) of this rank's contiguous block of draws (blocks differ by <= 1)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, extra = divmod(n_draw, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(x, n_draw=None, rank=None, world=None):
    """slice the leading (draw) dimension of a tensor / pytree of tensors for this rank"""
    if isinstance(x, dict):
        return {k: shard(v, n_draw, rank, world) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(shard(v, n_draw, rank, world) for v in x)
    n = x.shape[0] if n_draw is None else n_draw
    lo, hi = shard_bounds(n, rank, world)
    return x[lo:hi]


def gather_loglike(local, n_draw, group=None):
    """every rank receives the full (n_draw,) vector: each fills its own slice of a
    zero vector and ONE all-reduce(SUM) completes it (equivalent to an all-gather,
    but valid for ragged shards)."""
    out = torch.zeros(n_draw, dtype=local.dtype, device=local.device)
    lo, hi = shard_bounds(n_draw)
    if hi - lo != local.shape[0]:
        raise ValueError(f"rank owns draws [{lo},{hi}) but got {local.shape[0]} values")
    out[lo:hi] = local.detach()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def sharded_log_likelihood(loglike_fn, params, n_draw):
    """Evaluate ``loglike_fn(params_shard) -> (n_local,)`` on this rank's draws and
    return (full log-likelihood vector on every rank, local log-likelihoods with
    their autograd graph).  Gradients never leave the owning rank."""
    local_params = shard(params, n_draw)
    local = loglike_fn(local_params)
    return gather_loglike(local, n_draw), local
