"""Data-parallel helpers: one process per GPU, batch-sharded, RCCL ("nccl" backend) over xGMI.

The reference has no distributed code; this layer is new.  Convention used throughout: every rank
evaluates the SAME global loss value, built from global statistics that are obtained with
`dist_sum` (a differentiable all-reduce).  Each rank's backward then yields only the part of the
gradient that flows through its own shard, so parameter gradients are combined with an all-reduce
SUM (never an average) over one flat bucket per optimiser group.
"""
import os
from typing import Iterable

import torch
import torch.distributed as dist

# Testing aid: run every collective even in a group of one rank (they are identities there), so the
# data-parallel code path -- and its capture into HIP graphs -- can be exercised on a single GPU.
FORCE_COLLECTIVES = os.environ.get("TIMEGAN_FORCE_COLLECTIVES") == "1"


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def multi_rank(group=None) -> bool:
    """True when collectives have to run: more than one rank (or FORCE_COLLECTIVES with a live group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _staged(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: a device tensor handed to it is staged by the backend itself, one transfer
    per call and -- when several ranks share one GPU, as in rehearsals of the multi-rank flow on a one-GPU
    box -- prone to stalls.  Device tensors are therefore staged through the host here, explicitly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    """In-place SUM over the ranks of `group` (RCCL takes the device tensor as it is)."""
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class _DistSum(torch.autograd.Function):
    """S = sum over ranks of s_r.  Contract: every rank computes the SAME function of S afterwards
    (replicated arithmetic only), so dL/ds_r = dL/dS is available locally and the backward is the
    identity.  Feeding S into rank-specific arithmetic (e.g. subtracting it from local rows) breaks
    that contract; rebuild such terms from all-reduced raw sums instead."""

    @staticmethod
    def forward(ctx, x, group):
        out = x.clone()
        _all_reduce_sum(out, group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def dist_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    if not multi_rank(group):
        return x
    return _DistSum.apply(x, group)


def all_reduce_grads(modules: Iterable[torch.nn.Module], group=None, arena=None) -> int:
    """Sum the gradients of `modules` over all ranks.  Returns the number of floats exchanged.
    With `arena` (the deferred.GradArena the backward pass carved its gradient buffers from) the gradients that
    live in it -- on the GPU path all of them -- are summed by ONE in-place collective over the arena's used
    part: nothing is concatenated or copied back.  Gradients outside it (torch's own backward functions) go
    through one flat bucket: cat -> all-reduce -> one multi-tensor copy back."""
    if not multi_rank(group):
        return 0
    params = [p for m in modules for p in m.parameters() if p.grad is not None]
    if not params:
        return 0
    count = 0
    if arena is not None and arena.n > 0 and any(arena.owns(p.grad) for p in params):
        flat = arena.used()
        _all_reduce_sum(flat, group)
        count += int(flat.numel())
        params = [p for p in params if not arena.owns(p.grad)]
        if not params:
            return count
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    _all_reduce_sum(flat, group)
    views, off = [], 0
    for p in params:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p.grad))
        off += n
    torch._foreach_copy_([p.grad for p in params], views)    # one multi-tensor copy, not one per parameter
    return count + int(flat.numel())


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0, group=None) -> None:
    """Parameters and buffers of `modules` from rank `src` to every rank: ONE flat buffer (one collective)
    per network and dtype, not one broadcast per tensor."""
    if not multi_rank(group):
        return
    for m in modules:
        by_dtype = {}
        for t in list(m.parameters()) + list(m.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t.data)
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            if _staged(flat, group):
                c = flat.cpu()
                dist.broadcast(c, src=src, group=group)
                flat.copy_(c)
            else:
                dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


def shard_batch(x: torch.Tensor, group=None) -> torch.Tensor:
    """Contiguous shard of a global batch for this rank (sizes differ by at most one)."""
    w, r = world_size(group), rank(group)
    if w == 1:
        return x
    n = x.shape[0]
    base, rem = divmod(n, w)
    start = r * base + min(r, rem)
    return x[start:start + base + (1 if r < rem else 0)]


def shard_bounds(n: int, group=None):
    """(start, stop) of this rank's contiguous shard of n rows (the shards shard_batch cuts)."""
    w, r = world_size(group), rank(group)
    base, rem = divmod(n, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


def broadcast_int(value: int, device, src: int = 0, group=None) -> int:
    """`value` of rank `src` on every rank (a seed all ranks must share)."""
    if not multi_rank(group):
        return int(value)
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def equal_shard_extent(n_local: int, group=None):
    """(start, stop, total) of this rank's rows when every rank holds n_local rows: no communication."""
    w, r = world_size(group), rank(group)
    return r * n_local, (r + 1) * n_local, w * n_local


def shard_extent(n_local: int, device, group=None):
    """(start, stop, total) of this rank's rows inside the concatenation of all ranks' shards, from the
    shard sizes themselves (one small all-gather): works for any split, even or not."""
    if not multi_rank(group):
        return 0, n_local, n_local
    w, r = world_size(group), rank(group)
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(w)]
    dist.all_gather(parts, mine, group=group)
    sizes = [int(p.item()) for p in parts]
    start = sum(sizes[:r])
    return start, start + n_local, sum(sizes)


def all_gather_rows(x_local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Inverse of shard_batch: every rank passes its shard of an n_total-row tensor and receives the
    whole tensor.  ONE all-gather; shards that are one row shorter travel padded."""
    if not multi_rank(group):
        return x_local
    w = world_size(group)
    base, rem = divmod(n_total, w)
    cap = base + (1 if rem else 0)
    send = x_local.new_zeros((cap,) + tuple(x_local.shape[1:]))
    send[:x_local.shape[0]] = x_local
    parts = [torch.empty_like(send) for _ in range(w)]
    dist.all_gather(parts, send, group=group)
    return torch.cat([p[:base + (1 if r < rem else 0)] for r, p in enumerate(parts)])
