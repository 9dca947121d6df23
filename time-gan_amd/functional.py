"""Host side of the GRU operator: tensors -> pointers -> libtgru.so, wrapped for autograd.

The operator replaces `_VF.gru(...)` as reached from nn.GRU.forward
(torch/nn/modules/rnn.py:1427-1437), i.e. the call the reference makes at
outdated_files/TimeGANv1.py:61,139,220,299,376.
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib, deferred
from ._lib import BackwardArgs, ForwardArgs, LAYOUT_BTF, LAYOUT_PACKED, LAYOUT_TBF, check, lib, on_device


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _addr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _alloc_grads(weights: Sequence[torch.Tensor], queue=None):
    """One flat buffer for the gradients of `weights` and a view of it per weight.  Autograd receives the
    views (no one else references them, so AccumulateGrad adopts them without copying -- a copy taken
    before a deferred reduction has run would copy nothing useful); whoever must keep the memory alive
    holds the flat buffer.  `queue`: the pass's deferred-reduction queue when these are the weights' first
    gradients of the pass; the buffer then comes from the queue's gradient arena, if it has one.  Every view
    starts on a 16-byte boundary."""
    offs, n = [], 0
    for w in weights:
        offs.append(n)
        n += (w.numel() + 3) // 4 * 4
    dev = weights[0].device
    flat = queue.grad_buffer(n, dev) if queue is not None else torch.empty(n, device=dev, dtype=torch.float32)
    return flat, [flat[o:o + w.numel()].view(w.shape) for o, w in zip(offs, weights)]


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def _dense16(t: torch.Tensor) -> torch.Tensor:
    """`t` as a dense tensor whose storage starts on a 16-byte boundary, which the library requires of every
    buffer.  A contiguous slice of a larger tensor (x[1:], a rank's shard of a batch) is dense but may start
    anywhere: it is copied then."""
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.TgruError(
            f"{what} is on {t.device}: the tgru GRU only runs on a gfx950 GPU and has no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.TgruError(f"{what} must be float32, got {t.dtype}")


def _split_weights(flat: Sequence[torch.Tensor], L: int):
    assert len(flat) == 4 * L, "expected weight_ih, weight_hh, bias_ih, bias_hh per layer"
    return flat[0::4], flat[1::4], flat[2::4], flat[3::4]


def _dims(x: torch.Tensor, layout: int, B: int, T: int, weights: Sequence[torch.Tensor]):
    L = len(weights) // 4
    H = weights[1].shape[1]
    In = weights[0].shape[1]
    if layout == LAYOUT_TBF:
        assert tuple(x.shape) == (T, B, In), f"x {tuple(x.shape)} is not (T,B,In)=({T},{B},{In})"
    elif layout == LAYOUT_BTF:
        assert tuple(x.shape) == (B, T, In), f"x {tuple(x.shape)} is not (B,T,In)=({B},{T},{In})"
    else:
        assert x.dim() == 2 and x.shape[1] == In, f"packed x {tuple(x.shape)} is not (rows,{In})"
    return In, H, L


class _GRUFunction(torch.autograd.Function):
    """y, h_n = GRU(x, h0) with lengths; saves the library's reserve buffer for backward."""

    @staticmethod
    def forward(ctx, x, h0, lengths, toff, B, T, layout, grad_mode, *weights):
        _require_gpu(x, "input")
        In, H, L = _dims(x, layout, B, T, weights)
        x = _dense16(x)
        weights = tuple(_dense16(w) for w in weights)
        for w in weights:
            _require_gpu(w, "weight")
        h0c = None if h0 is None else _dense16(h0)
        w_ih, w_hh, b_ih, b_hh = _split_weights(weights, L)
        dev = x.device
        # `grad_mode` is torch.is_grad_enabled() of the CALLER (inside forward() it is always off);
        # needs_input_grad alone is also true under torch.no_grad(), where nothing must be saved
        need_grad = grad_mode and any(ctx.needs_input_grad)
        with on_device(dev):
            lb = lib()
            y = torch.empty(x.shape[:-1] + (H,), device=dev, dtype=torch.float32)
            if layout == LAYOUT_PACKED:
                pass   # every packed row is an active step, all rows get written
            h_n = torch.empty((L, B, H), device=dev, dtype=torch.float32)
            ws = torch.empty(lb.tgru_forward_workspace_bytes(B, T, In, H, L) // 4, device=dev,
                             dtype=torch.float32)
            reserve = None
            if need_grad:
                reserve = torch.empty(lb.tgru_reserve_bytes(B, T, In, H, L, layout) // 4, device=dev,
                                      dtype=torch.float32)
            rc = lb.tgru_forward(_ptr(x), _ptr(h0c), _ptr(lengths), _ptr(toff),
                                 _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh),
                                 _ptr(y), _ptr(h_n), _ptr(reserve), _ptr(ws),
                                 B, T, In, H, L, layout, _stream(dev))
            check(rc, "tgru_forward")
        ctx.dims = (B, T, In, H, L, layout)
        ctx.has_h0 = h0 is not None
        ctx.set_materialize_grads(False)   # an unused h_n (or y) arrives in backward as None, not as zeros
        ctx.save_for_backward(x, h0c, y, lengths, toff, reserve, *weights)
        return y, h_n

    @staticmethod
    def backward(ctx, dy, dh_n):
        B, T, In, H, L, layout = ctx.dims
        x, h0, y, lengths, toff, reserve, *weights = ctx.saved_tensors
        dev = x.device
        dy = torch.zeros_like(y) if dy is None else _dense16(dy)   # None: only h_n was used
        dh_n = None if dh_n is None else _dense16(dh_n)
        with on_device(dev):
            lb = lib()
            ws = torch.empty(lb.tgru_backward_workspace_bytes(B, T, In, H, L) // 4, device=dev,
                             dtype=torch.float32)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dh0 = torch.empty_like(h0) if (ctx.has_h0 and ctx.needs_input_grad[1]) else None
            # no weight of this call wants a gradient (a network the step does not update): the library is
            # told so (NULL gradient arrays) and forms dx / dh0 only
            want_w = any(ctx.needs_input_grad[8:])
            queue = deferred.active()
            defer = want_w and queue is not None and queue.may_defer(weights)
            flat, grads = _alloc_grads(weights, queue if defer else None) if want_w else (None, [None] * len(weights))
            arrays = [_ptr_array(g) for g in _split_weights(weights, L)]
            if want_w:
                arrays += [_ptr_array(g) for g in _split_weights(grads, L)]
            cast = [ctypes.cast(a, ctypes.c_void_p) for a in arrays] + ([] if want_w else [None] * 4)
            args = BackwardArgs(_addr(dy), _addr(dh_n), _addr(x), _addr(h0), _addr(y), _addr(lengths), _addr(toff),
                                *cast[:4], _addr(reserve), _addr(ws), _addr(dx), _addr(dh0), *cast[4:],
                                B, T, In, H, L, layout, _lib.DEFER_REDUCE if defer else 0)
            rc = lb.tgru_backward(args.dy, args.dh_n, args.x, args.h0, args.y, args.lengths, args.toff,
                                  args.w_ih, args.w_hh, args.b_ih, args.b_hh, args.reserve, args.workspace,
                                  args.dx, args.dh0, args.dw_ih, args.dw_hh, args.db_ih, args.db_hh,
                                  B, T, In, H, L, layout, args.accumulate, _stream(dev))
            check(rc, "tgru_backward")
            if defer:
                queue.add(lb.tgru_backward_reduce_jobs, args, [ws, arrays, flat], weights, dev)
        return (dx, dh0, None, None, None, None, None, None, *grads)


class _GRUPairFunction(torch.autograd.Function):
    """(yA, h_nA, yB, h_nB) = two independent GRU calls through tgru_forward_pair / tgru_backward_pair:
    when both run on the one-sequence-per-workgroup kernels they share one launch forward and one
    backward (the two stacks run side by side on the GPU).  `specs` = ((B, T, layout), (B, T, layout), grad mode of the caller);
    `tensors` = x, h0, lengths, toff, *weights of call A, then the same of call B."""

    @staticmethod
    def forward(ctx, specs, *tensors):
        nw = (len(tensors) - 8) // 2
        calls = []
        for k in range(2):
            x, h0, lengths, toff, *weights = tensors[k * (4 + nw):(k + 1) * (4 + nw)]
            B, T, layout = specs[k]      # (specs[2] is the caller's grad mode)
            _require_gpu(x, "input")
            In, H, L = _dims(x, layout, B, T, weights)
            calls.append(dict(x=_dense16(x), h0=None if h0 is None else _dense16(h0), lengths=lengths,
                              toff=toff, weights=tuple(_dense16(w) for w in weights), B=B, T=T, In=In, H=H,
                              L=L, layout=layout))
        dev = calls[0]["x"].device
        need_grad = specs[2] and any(ctx.needs_input_grad)     # specs[2]: the caller's grad mode
        keep, args = [], []
        with on_device(dev):
            lb = lib()
            for k, c in enumerate(calls):
                B, T, In, H, L, layout = (c[k_] for k_ in ("B", "T", "In", "H", "L", "layout"))
                f32 = dict(device=dev, dtype=torch.float32)
                # the reserve (gate planes, layer outputs) only for a call that will see a backward pass
                need_k = need_grad and any(ctx.needs_input_grad[1 + k * (4 + nw):1 + (k + 1) * (4 + nw)])
                c["y"] = torch.empty(c["x"].shape[:-1] + (H,), **f32)
                c["h_n"] = torch.empty((L, B, H), **f32)
                ws = torch.empty(lb.tgru_forward_workspace_bytes(B, T, In, H, L) // 4, **f32)
                c["reserve"] = torch.empty(lb.tgru_reserve_bytes(B, T, In, H, L, layout) // 4, **f32) \
                    if need_k else None
                arrays = [_ptr_array(g) for g in _split_weights(c["weights"], L)]
                keep += arrays + [ws]
                args.append(ForwardArgs(_addr(c["x"]), _addr(c["h0"]), _addr(c["lengths"]), _addr(c["toff"]),
                                        *[ctypes.cast(a, ctypes.c_void_p) for a in arrays],
                                        _addr(c["y"]), _addr(c["h_n"]), _addr(c["reserve"]), _addr(ws),
                                        B, T, In, H, L, layout))
            check(lb.tgru_forward_pair(ctypes.byref(args[0]), ctypes.byref(args[1]), _stream(dev)),
                  "tgru_forward_pair")
        ctx.set_materialize_grads(False)
        ctx.meta = [tuple(c[k] for k in ("B", "T", "In", "H", "L", "layout")) + (c["h0"] is not None,)
                    for c in calls]
        ctx.nw = nw
        saved = []
        for c in calls:
            saved += [c["x"], c["h0"], c["y"], c["lengths"], c["toff"], c["reserve"], *c["weights"]]
        ctx.save_for_backward(*saved)
        # autograd takes every output of a Function to depend on every input; a call none of whose own inputs
        # wants a gradient (a network that is not updated, fed by data) must not drag a backward pass behind it
        for k in range(2):
            if not any(ctx.needs_input_grad[1 + k * (4 + nw):1 + (k + 1) * (4 + nw)]):
                ctx.mark_non_differentiable(calls[k]["y"], calls[k]["h_n"])
        return calls[0]["y"], calls[0]["h_n"], calls[1]["y"], calls[1]["h_n"]

    @staticmethod
    def backward(ctx, *douts):
        nw = ctx.nw
        saved = ctx.saved_tensors
        dev = saved[0].device
        keep, args, results = [], [], []
        # per call: does any weight want a gradient, does anything at all (a call that owes nothing is left out)
        want_w = [any(ctx.needs_input_grad[1 + k * (4 + nw) + 4:1 + (k + 1) * (4 + nw)]) for k in range(2)]
        live = [k for k in range(2)
                if want_w[k] or ctx.needs_input_grad[1 + k * (4 + nw)]
                or (ctx.meta[k][6] and ctx.needs_input_grad[2 + k * (4 + nw)])]
        all_w = [w for k in range(2) if want_w[k] for w in saved[k * (6 + nw) + 6:(k + 1) * (6 + nw)]]
        queue = deferred.active()
        defer = bool(all_w) and queue is not None and queue.may_defer(all_w)
        flag = _lib.DEFER_REDUCE if defer else 0
        with on_device(dev):
            lb = lib()
            for k in range(2):
                if k not in live:
                    results.append((None, None, [None] * nw))
                    continue
                x, h0, y, lengths, toff, reserve, *weights = saved[k * (6 + nw):(k + 1) * (6 + nw)]
                B, T, In, H, L, layout, has_h0 = ctx.meta[k]
                dy, dh_n = douts[2 * k], douts[2 * k + 1]
                dy = torch.zeros_like(y) if dy is None else _dense16(dy)
                dh_n = None if dh_n is None else _dense16(dh_n)
                base = 1 + k * (4 + nw)          # position of this call's x among the forward inputs
                ws = torch.empty(lb.tgru_backward_workspace_bytes(B, T, In, H, L) // 4, device=dev,
                                 dtype=torch.float32)
                dx = torch.empty_like(x) if ctx.needs_input_grad[base] else None
                dh0 = torch.empty_like(h0) if (has_h0 and ctx.needs_input_grad[base + 1]) else None
                flat, grads = _alloc_grads(weights, queue if defer else None) if want_w[k] else (None, [None] * nw)
                arrays = [_ptr_array(g) for g in _split_weights(weights, L)]
                if want_w[k]:
                    arrays += [_ptr_array(g) for g in _split_weights(grads, L)]
                keep += arrays + [ws, dy, dh_n, flat]
                cast = [ctypes.cast(a, ctypes.c_void_p) for a in arrays] + ([] if want_w[k] else [None] * 4)
                args.append(BackwardArgs(_addr(dy), _addr(dh_n), _addr(x), _addr(h0), _addr(y), _addr(lengths),
                                         _addr(toff), *cast[:4], _addr(reserve), _addr(ws), _addr(dx), _addr(dh0),
                                         *cast[4:], B, T, In, H, L, layout, flag))
                results.append((dx, dh0, grads))
            if len(args) == 2:
                check(lb.tgru_backward_pair(ctypes.byref(args[0]), ctypes.byref(args[1]), _stream(dev)),
                      "tgru_backward_pair")
            else:
                a = args[0]
                check(lb.tgru_backward(a.dy, a.dh_n, a.x, a.h0, a.y, a.lengths, a.toff, a.w_ih, a.w_hh, a.b_ih, a.b_hh,
                                       a.reserve, a.workspace, a.dx, a.dh0, a.dw_ih, a.dw_hh, a.db_ih, a.db_hh,
                                       a.B, a.T, a.In, a.H, a.L, a.layout, a.accumulate, _stream(dev)), "tgru_backward")
            if defer:
                for a in args:
                    queue.add(lb.tgru_backward_reduce_jobs, a, keep, all_w, dev)
        out = [None]
        for dx, dh0, grads in results:
            out += [dx, dh0, None, None, *grads]
        return tuple(out)


def gru_apply_pair(call_a, call_b):
    """Two independent differentiable GRU calls evaluated together.  Each call is a tuple
    (x, weights, B, T, layout, h0, lengths, toff) with gru_apply's meaning; the weight lists must
    have the same length (same number of layers).  Returns ((yA, h_nA), (yB, h_nB))."""
    specs, tensors = [], []
    for x, weights, B, T, layout, h0, lengths, toff in (call_a, call_b):
        specs.append((B, T, layout))
        tensors += [x, h0, lengths, toff, *weights]
    ya, ha, yb, hb = _GRUPairFunction.apply((specs[0], specs[1], torch.is_grad_enabled()), *tensors)
    return (ya, ha), (yb, hb)


_cpp_state = {"tried": False, "module": None}


def cpp_op():
    """The C++ autograd binding of the GRU (time-gan_amd/_tgru_torch.so, built by csrc_torch/build.sh: a torch
    extension over the same C ABI that reads the current HIP stream itself and runs its backward pass in
    autograd's thread without entering Python), or None: not built, switched off with TGRU_NO_CPP_OP=1, or
    another build of the library selected with TGRU_LIB_PATH (the extension is linked to the in-tree one)."""
    if not _cpp_state["tried"]:
        _cpp_state["tried"] = True
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tgru_torch.so")
        if (os.path.exists(path) and os.environ.get("TGRU_NO_CPP_OP") != "1" and not os.environ.get("TGRU_LIB_PATH")):
            lib()      # the C-ABI library first: a missing or broken libtgru.so must fail as loudly as ever
            spec = importlib.util.spec_from_file_location("_tgru_torch", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _cpp_state["module"] = mod
    return _cpp_state["module"]


def gru_apply(x: torch.Tensor, weights: Sequence[torch.Tensor], B: int, T: int, layout: int,
              h0: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
              toff: Optional[torch.Tensor] = None, cpp: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Differentiable multi-layer GRU.

    x in `layout` ((T,B,In), (B,T,In) or packed rows); weights flat in torch's order
    [weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0, weight_ih_l1, ...];
    lengths / toff: int32 device tensors (see include/tgru.h), None for full-length sequences.
    `cpp`: go through the C++ autograd binding when it is there (same C ABI, same results; its weight gradients
    are complete when backward returns, so callers that collect theirs in a deferred-reduction queue leave it off).
    Returns (y in the same layout, h_n (L,B,H)).
    """
    if cpp and x.is_cuda and deferred.active() is None:
        mod = cpp_op()
        if mod is not None:
            y, h_n = mod.gru_apply(x, h0, lengths, toff, B, T, layout, list(weights))
            return y, h_n
    return _GRUFunction.apply(x, h0, lengths, toff, B, T, layout, torch.is_grad_enabled(), *weights)


def gru_forward_backward_raw(x, weights, B, T, layout, dy, h0=None, dh_n=None, lengths=None, toff=None,
                             want_dx=True):
    """One forward + backward straight through the C ABI (no autograd graph); for benchmarks and tests.

    Returns dict(y, h_n, dx, grads).
    """
    _require_gpu(x, "input")
    In, H, L = _dims(x, layout, B, T, weights)
    w_ih, w_hh, b_ih, b_hh = _split_weights(weights, L)
    dev = x.device
    with on_device(dev):
        lb = lib()
        y = torch.empty(x.shape[:-1] + (H,), device=dev, dtype=torch.float32)
        h_n = torch.empty((L, B, H), device=dev, dtype=torch.float32)
        ws = torch.empty(max(lb.tgru_forward_workspace_bytes(B, T, In, H, L),
                             lb.tgru_backward_workspace_bytes(B, T, In, H, L)) // 4, device=dev,
                         dtype=torch.float32)
        reserve = torch.empty(lb.tgru_reserve_bytes(B, T, In, H, L, layout) // 4, device=dev,
                              dtype=torch.float32)
        s = _stream(dev)
        check(lb.tgru_forward(_ptr(x), _ptr(h0), _ptr(lengths), _ptr(toff),
                              _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh),
                              _ptr(y), _ptr(h_n), _ptr(reserve), _ptr(ws),
                              B, T, In, H, L, layout, s), "tgru_forward")
        dx = torch.empty_like(x) if want_dx else None
        dh0 = torch.empty_like(h0) if h0 is not None else None
        grads = [torch.empty_like(w) for w in weights]
        gw_ih, gw_hh, gb_ih, gb_hh = _split_weights(grads, L)
        check(lb.tgru_backward(_ptr(dy), _ptr(dh_n), _ptr(x), _ptr(h0), _ptr(y), _ptr(lengths), _ptr(toff),
                               _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh),
                               _ptr(reserve), _ptr(ws), _ptr(dx), _ptr(dh0),
                               _ptr_array(gw_ih), _ptr_array(gw_hh), _ptr_array(gb_ih), _ptr_array(gb_hh),
                               B, T, In, H, L, layout, 0, s), "tgru_backward")
    return {"y": y, "h_n": h_n, "dx": dx, "dh0": dh0, "grads": grads}


def uniform_(out: torch.Tensor, seed: int, stream: int, first: int = 0) -> torch.Tensor:
    """Fills the dense fp32 GPU tensor `out` with elements [first, first + out.numel()) of noise stream `stream`
    under `seed`, U[0, 1) (tgru_uniform of include/tgru.h: a pure function of (seed, stream, element), so shards of
    one tensor can be drawn independently)."""
    _require_gpu(out, "out")
    assert out.is_contiguous()
    with on_device(out.device):
        check(lib().tgru_uniform(_ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(stream) & (2 ** 64 - 1),
                                 int(first), _stream(out.device)), "tgru_uniform")
    return out


def input_projection(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a @ w.T + bias through the library's fp32 MFMA GEMM (no autograd)."""
    _require_gpu(a, "a")
    _require_gpu(w, "w")
    a = a.contiguous()
    w = w.contiguous()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    with on_device(a.device):
        check(lib().tgru_input_projection(_ptr(a), _ptr(w), _ptr(bias), _ptr(out), M, N, K,
                                          _stream(a.device)), "tgru_input_projection")
    return out


def lengths_to_device(lengths, B: int, T: int, device) -> Optional[torch.Tensor]:
    """Validate `lengths` the way pack_padded_sequence(enforce_sorted=True) does and move to device.

    Returns None when every sequence has length T (the kernels' uniform fast path).
    """
    if lengths is None:
        return None
    if isinstance(lengths, torch.Tensor):
        lens = lengths.detach().to("cpu").to(torch.int64).tolist()
    else:
        lens = [int(v) for v in lengths]
    if len(lens) != B:
        raise RuntimeError(f"Expected `len(lengths)` to be equal to batch_size, but got {len(lens)} "
                           f"(batch_size={B})")
    if lens.count(T) == B:      # the common case (every window has full length), checked at C speed
        return None
    if min(lens) <= 0:
        raise RuntimeError("Length of all samples has to be greater than 0, but found an element in "
                           "'lengths' that is <= 0")
    if any(lens[i] < lens[i + 1] for i in range(B - 1)):
        raise RuntimeError("`lengths` array must be sorted in decreasing order when "
                           "`enforce_sorted` is True. You can pass `enforce_sorted=False` to "
                           "pack_padded_sequence and/or pad_packed_sequence to sidestep this requirement "
                           "if you do not need ONNX exportability.")
    if max(lens) > T:
        raise RuntimeError(f"a length ({max(lens)}) exceeds the sequence dimension ({T})")
    if min(lens) == T:
        return None
    return torch.tensor(lens, dtype=torch.int32, device=device)


class GRURunner:
    """Pre-allocated buffers + direct C-ABI calls for one fixed problem shape.

    No allocation and no autograd bookkeeping per step: `forward()` then `backward(dy)` issue only
    the library's kernel launches on the current stream.  All parameter gradients live in one flat
    buffer (`flat_grads`) so a data-parallel caller can all-reduce them with one collective.
    """

    def __init__(self, B: int, T: int, In: int, H: int, L: int, layout: int, device, want_dx: bool = True):
        self.dims = (B, T, In, H, L, layout)
        self.device = torch.device(device)
        lb = lib()
        f32 = dict(device=self.device, dtype=torch.float32)
        shape = {LAYOUT_TBF: (T, B), LAYOUT_BTF: (B, T)}[layout]
        self.y = torch.empty(shape + (H,), **f32)
        self.h_n = torch.empty((L, B, H), **f32)
        self.dx = torch.empty(shape + (In,), **f32) if want_dx else None
        self.ws = torch.empty(max(lb.tgru_forward_workspace_bytes(B, T, In, H, L),
                                  lb.tgru_backward_workspace_bytes(B, T, In, H, L)) // 4, **f32)
        self.reserve = torch.empty(lb.tgru_reserve_bytes(B, T, In, H, L, layout) // 4, **f32)
        shapes = []
        for l in range(L):
            in_l = In if l == 0 else H
            shapes += [(3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)]
        # keep every gradient 16-byte aligned inside the flat buffer
        offs, n = [], 0
        for s in shapes:
            offs.append(n)
            n += (int(torch.Size(s).numel()) + 3) // 4 * 4
        self.flat_grads = torch.zeros(n, **f32)
        self.grads = [self.flat_grads[o:o + torch.Size(s).numel()].view(s) for o, s in zip(offs, shapes)]
        self._gptr = [_ptr_array(g) for g in _split_weights(self.grads, L)]
        self.weights = None
        self._wptr = None

    def set_weights(self, weights: Sequence[torch.Tensor]):
        L = self.dims[4]
        self.weights = [w.contiguous() for w in weights]
        self._wptr = [_ptr_array(w) for w in _split_weights(self.weights, L)]

    def forward(self, x: torch.Tensor, lengths=None):
        B, T, In, H, L, layout = self.dims
        self.x = x
        self.lengths = lengths
        check(lib().tgru_forward(_ptr(x), None, _ptr(lengths), None, *self._wptr,
                                 _ptr(self.y), _ptr(self.h_n), _ptr(self.reserve), _ptr(self.ws),
                                 B, T, In, H, L, layout, _stream(self.device)), "tgru_forward")
        return self.y

    def backward(self, dy: torch.Tensor):
        B, T, In, H, L, layout = self.dims
        check(lib().tgru_backward(_ptr(dy), None, _ptr(self.x), None, _ptr(self.y), _ptr(self.lengths), None,
                                  *self._wptr, _ptr(self.reserve), _ptr(self.ws), _ptr(self.dx), None,
                                  *self._gptr, B, T, In, H, L, layout, 0, _stream(self.device)),
              "tgru_backward")
        return self.dx
