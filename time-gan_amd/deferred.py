"""Deferred gradient reductions: one launch for the partial-gradient sums of a whole backward pass.

Inside `with deferred_reductions():` the backward functions of the GRU and head ops ask the library
to leave their per-workgroup partial parameter gradients unreduced (TGRU_DEFER_REDUCE) and queue the
owed reductions; leaving the block runs all of them in ONE kernel (tgru_reduce_jobs).  Until then the
parameter-gradient tensors those backward calls returned hold no valid data, which is fine for
autograd as long as nothing reads them: AccumulateGrad only stores a first gradient.  A second
gradient for the same parameter is ADDED to the first, so only the first gradient a parameter receives
in the pass is deferred: a backward call that meets a parameter seen before flushes the queue if that
parameter is still pending and reduces its own gradients immediately.

The trainer wraps every `loss.backward()` in this block and leaves it before gradients are all-reduced
or handed to the optimiser.  Outside the block nothing is deferred.
"""
import ctypes
from contextlib import contextmanager

import torch

from ._lib import MAX_JOBS_PER_CALL, ReduceJob, check, lib, on_device

_active = None


_CAPACITY = 512     # jobs a queue can hold before it has to flush on its own


class GradArena:
    """One flat device buffer that the backward functions of a pass carve their parameter-gradient buffers from
    (a bump allocator, 16-byte granules).  A data-parallel trainer keeps one per optimiser group and passes it to
    `deferred_reductions(arena=...)`: every parameter's FIRST gradient of the pass -- the tensor autograd stores as
    `.grad` without copying -- is then a view of this buffer, later gradients of the same parameter are added into
    it in place, and the group's gradient exchange is ONE in-place all-reduce of `used()` with nothing to
    concatenate or copy back (`parallel.all_reduce_grads`).  The buffer is reused by the next pass: gradients of
    an earlier pass that nobody has zeroed are overwritten, which is when the reference zeroes them anyway
    (outdated_files/TimeGANv1.py:765-769)."""

    def __init__(self, capacity: int, device):
        self.buf = torch.empty(max(int(capacity), 4), device=device, dtype=torch.float32)
        self.n = 0

    def reset(self):
        self.n = 0

    def take(self, count: int):
        """`count` floats of the arena, or None when it is full (the caller allocates as usual)."""
        end = self.n + (count + 3) // 4 * 4
        if end > self.buf.numel():
            return None
        out = self.buf[self.n:self.n + count]
        self.n = end
        return out

    def used(self) -> torch.Tensor:
        return self.buf[:self.n]

    def owns(self, t: torch.Tensor) -> bool:
        lo = self.buf.data_ptr()
        return t.is_contiguous() and lo <= t.data_ptr() and t.data_ptr() + 4 * t.numel() <= lo + 4 * self.n


class _Queue:
    """The owed reductions of one backward pass.  The library writes the jobs of each call straight into
    one preallocated job array (no Python object per job)."""

    def __init__(self, arena=None):
        self.arena = arena        # GradArena the pass's first gradients are carved from, or None
        self.jobs = (ReduceJob * _CAPACITY)()
        self.count = 0
        self.keep, self.device = [], None
        self.pending = set()      # parameters whose (first) gradient is still unreduced
        self.seen = set()         # parameters that have received a gradient in this pass

    def may_defer(self, weights) -> bool:
        """Called by a backward function for the parameters it is about to produce gradients for.
        Only a parameter's FIRST gradient of the pass may be deferred (autograd just stores it); a later
        one is added to it, so the first must be complete by then -- pending work is flushed -- and the
        later one is reduced at once.  The same holds for a parameter that already HAS a gradient when the
        pass starts (gradient accumulation over several passes, zero_grad(set_to_none=False)): autograd adds
        to it on arrival.  And with create_graph=True (grad mode on inside backward) autograd clones what it
        receives at once.  Neither case is deferred."""
        ptrs = {w.data_ptr() for w in weights}
        if len(ptrs) != len(weights) or not self.pending.isdisjoint(ptrs):
            self.flush()
        first = self.seen.isdisjoint(ptrs) and len(ptrs) == len(weights)
        self.seen |= ptrs
        if first and (torch.is_grad_enabled() or any(w.grad is not None for w in weights)):
            self.flush()
            return False
        return first

    def grad_buffer(self, count: int, device) -> torch.Tensor:
        """Flat buffer for the gradients a backward call is about to return for parameters that receive their
        first gradient of the pass (may_defer() said so): from the arena when there is one with room."""
        buf = self.arena.take(count) if (self.arena is not None and self.arena.buf.device == device) else None
        return buf if buf is not None else torch.empty(count, device=device, dtype=torch.float32)

    def add(self, fn, args_struct, keep, weights, device):
        """Queue what the deferred backward call described by `args_struct` still owes
        (fn = the library's tgru_backward_reduce_jobs or tgru_head_reduce_jobs)."""
        if self.count + MAX_JOBS_PER_CALL > _CAPACITY:
            self.flush()
        slot = ctypes.cast(ctypes.byref(self.jobs, self.count * ctypes.sizeof(ReduceJob)), ctypes.POINTER(ReduceJob))
        n = fn(ctypes.byref(args_struct), slot, MAX_JOBS_PER_CALL)
        if n < 0:
            check(n, "reduce_jobs")
        self.count += n
        self.keep.append(keep)          # workspaces and gradient buffers stay alive until the flush
        self.pending.update(w.data_ptr() for w in weights)
        self.device = device

    def flush(self):
        if self.count:
            with on_device(self.device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                check(lib().tgru_reduce_jobs(self.jobs, self.count, stream), "tgru_reduce_jobs")
        self.count, self.keep, self.pending = 0, [], set()


def active():
    """The queue of the enclosing deferred_reductions() block, or None."""
    return _active


@contextmanager
def deferred_reductions(arena=None):
    """`arena` (a GradArena, reset here): where this pass's first parameter gradients are placed."""
    global _active
    if _active is not None:          # nested: the outer block owns the queue
        yield _active
        return
    if arena is not None:
        arena.reset()
    q = _Queue(arena)
    _active = q
    try:
        yield q
    finally:
        _active = None
        q.flush()
