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


class _Queue:
    """The owed reductions of one backward pass.  The library writes the jobs of each call straight into
    one preallocated job array (no Python object per job)."""

    def __init__(self):
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
def deferred_reductions():
    global _active
    if _active is not None:          # nested: the outer block owns the queue
        yield _active
        return
    q = _Queue()
    _active = q
    try:
        yield q
    finally:
        _active = None
        q.flush()
