"""The TimeGAN loss expressions as fused GPU ops (libtgru.so: tgru_loss_*).

Each `kind` is one of the loss builders of outdated_files/TimeGANv1.py:410-553 from the network outputs
on: two or three kernels forward and one backward instead of the 30-100 that the composition of
mse_loss / binary_cross_entropy_with_logits / mean / std / abs / sqrt takes.  With a process group of
more than one rank the partial sums are all-reduced between the kernels, so every rank gets the loss
of the concatenated batch and, from backward, its gradient with respect to the local rows (the same
convention as `parallel.dist_sum`: gradients of parameters are then summed over ranks).
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from ._lib import check, lib, on_device
from .parallel import _all_reduce_sum, multi_rank

EMBED_SIDE, SUPERVISOR, EMBED_MAIN, DISCRIMINATOR, GENERATOR = range(5)

_PtrArray6 = ctypes.c_void_p * 6
_PtrArray3 = ctypes.c_void_p * 3


def _ptrs(tensors: Sequence[Optional[torch.Tensor]], n: int):
    arr = (ctypes.c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def usable(*tensors: torch.Tensor) -> bool:
    """True when the fused ops cover these tensors: fp32 on a GPU."""
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


class _LossFunction(torch.autograd.Function):
    """(out0, out1, out2) = loss `kind` of `tensors`; see include/tgru.h for the operands per kind."""

    @staticmethod
    def forward(ctx, kind, dims, gamma, group, *tensors):
        B, T, H, F = dims
        tensors = tuple(t.contiguous() for t in tensors)
        dev = tensors[0].device
        ctx.set_materialize_grads(False)
        with on_device(dev):
            lb = lib()
            lay = (ctypes.c_int64 * 4)()
            check(lb.tgru_loss_layout(kind, T, F, lay), "tgru_loss_layout")
            n_a, n_b, off_out, total = (int(v) for v in lay)
            buf = torch.empty(total, device=dev, dtype=torch.float32)
            inp = _ptrs(tensors, 6)
            st = _stream(dev)
            bp = ctypes.c_void_p(buf.data_ptr())
            if not multi_rank(group):
                check(lb.tgru_loss_partial(kind, 0, inp, B, T, H, F, bp, st), "tgru_loss_partial")
            else:
                check(lb.tgru_loss_partial(kind, 1, inp, B, T, H, F, bp, st), "tgru_loss_partial")
                _all_reduce_sum(buf[:n_a], group)
                if n_b:
                    check(lb.tgru_loss_partial(kind, 2, inp, B, T, H, F, bp, _stream(dev)), "tgru_loss_partial")
                    _all_reduce_sum(buf[n_a:n_a + n_b], group)
            check(lb.tgru_loss_final(kind, B, T, H, F, float(gamma), bp, _stream(dev)), "tgru_loss_final")
        ctx.meta = (kind, dims, float(gamma))
        ctx.save_for_backward(buf, *tensors)
        out = buf[off_out:off_out + 3]
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        kind, (B, T, H, F), gamma = ctx.meta
        buf, *tensors = ctx.saved_tensors
        dev = buf.device
        grads = [torch.empty_like(t) if ctx.needs_input_grad[4 + i] else None for i, t in enumerate(tensors)]
        ups = [None if g is None else g.contiguous() for g in (g0, g1, g2)]
        with on_device(dev):
            check(lib().tgru_loss_backward(kind, _ptrs(tensors, 6), _ptrs(grads, 6), _ptrs(ups, 3), B, T, H, F,
                                           gamma, ctypes.c_void_p(buf.data_ptr()), _stream(dev)),
                  "tgru_loss_backward")
        return (None, None, None, None, *grads)


def fused_loss(kind: int, tensors: Sequence[torch.Tensor], B: int, T: int, H: int, F: int,
               gamma: float = 1.0, group=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return _LossFunction.apply(kind, (B, T, H, F), gamma, group, *tensors)
