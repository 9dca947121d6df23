"""Fused dense head of the TimeGAN networks on the GPU (libtgru.so: tgru_head_forward/backward).

    y = [LayerNorm] -> Linear -> [LayerNorm -> ReLU -> Linear]    (per row; in / out <= 128 wide, middle <= 192)

The op takes the parameters of ordinary torch modules (nn.LayerNorm, nn.Linear), so the networks
keep torch's module structure and state_dict keys (outdated_files/TimeGANv1.py:39-50,119-128,
198-207,276-287,355-357); only their forward goes through one kernel instead of 5-8, and their
backward through two instead of about twenty.
"""
import ctypes
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, deferred
from ._lib import HeadArgs, check, lib, on_device

MAX_IO_WIDTH = 32       # up to here (middle: MAX_MID_WIDTH) the kernels keep the weights in registers and
MAX_MID_WIDTH = 64      # two heads can share a launch; wider heads (tgru_head_supported) keep them in LDS


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _addr(t):
    return None if t is None else t.data_ptr()


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _HeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, g0, b0, w1, c1, g1, b1, w2, c2):
        din = x.shape[-1]
        dmid = w1.shape[0]
        dout = dmid if w2 is None else w2.shape[0]
        x2 = x.contiguous().view(-1, din)
        M = x2.shape[0]
        params = [None if p is None else p.contiguous() for p in (g0, b0, w1, c1, g1, b1, w2, c2)]
        y = torch.empty(x.shape[:-1] + (dout,), device=x.device, dtype=torch.float32)
        with on_device(x.device):
            check(lib().tgru_head_forward(_ptr(x2), _ptr(y), M, din, dmid, dout, *[_ptr(p) for p in params],
                                          float(eps), _stream(x.device)), "tgru_head_forward")
        ctx.dims = (M, din, dmid, dout, float(eps))
        ctx.save_for_backward(x2, *[p for p in params if p is not None])
        ctx.present = [p is not None for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        M, din, dmid, dout, eps = ctx.dims
        saved = list(ctx.saved_tensors)
        x2 = saved.pop(0)
        params = [saved.pop(0) if pres else None for pres in ctx.present]
        dev = x2.device
        dy2 = dy.contiguous().view(-1, dout)
        with on_device(dev):
            lb = lib()
            n = lb.tgru_head_grad_floats(din, dmid, dout)
            # no parameter wants a gradient (a network the step does not update): dx only
            want_p = any(ctx.needs_input_grad[2:])
            live = [p for p in params if p is not None]
            queue = deferred.active()
            defer = want_p and queue is not None and queue.may_defer(live)
            # (first gradients of the pass: from the pass's gradient arena when it has one)
            dparams = None
            if want_p:
                dparams = queue.grad_buffer(n, dev) if defer else torch.empty(n, device=dev, dtype=torch.float32)
            ws = torch.empty(lb.tgru_head_backward_workspace_bytes(M, din, dmid, dout) // 4, device=dev,
                             dtype=torch.float32)
            dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
            flag = _lib.DEFER_REDUCE if defer else 0
            check(lb.tgru_head_backward(_ptr(x2), _ptr(dy2), _ptr(dx), M, din, dmid, dout,
                                        *[_ptr(p) for p in params], eps, _ptr(dparams), _ptr(ws), flag,
                                        _stream(dev)), "tgru_head_backward")
            if defer:
                args = HeadArgs(_addr(x2), None, _addr(dy2), _addr(dx), M, din, dmid, dout,
                                *[_addr(p) for p in params], eps, _addr(dparams), _addr(ws), flag)
                queue.add(lb.tgru_head_reduce_jobs, args, [ws, dparams, dy2], live, dev)
        grads = _split_grads(dparams, din, dmid, dout, params) if want_p else [None] * 8
        if dx is not None:
            dx = dx.view(dy.shape[:-1] + (din,))
        return (dx, None, *grads)


def supported(x: torch.Tensor, lin1: nn.Linear, lin2: Optional[nn.Linear], lns=()) -> bool:
    """True when the fused kernels cover this head: GPU fp32 rows, widths the library takes
    (tgru_head_supported: input / output up to 128, the middle between the two Linears up to 192), biases
    present."""
    if not x.is_cuda or x.dtype != torch.float32 or x.numel() == 0:
        return False      # (an empty shard keeps the torch modules, which return an empty result)
    if lin1.bias is None or (lin2 is not None and lin2.bias is None):
        return False
    if any(ln is not None and (ln.weight is None or ln.bias is None) for ln in lns):
        return False      # LayerNorm(elementwise_affine=False): the fused kernels expect gamma and beta
    if any(p.dtype != torch.float32 for p in (lin1.weight, lin1.bias)):
        return False
    dout = lin1.out_features if lin2 is None else lin2.out_features
    return bool(lib().tgru_head_supported(lin1.in_features, lin1.out_features, dout, int(lin2 is not None)))


def head_apply(x: torch.Tensor, ln0: Optional[nn.LayerNorm], lin1: nn.Linear,
               ln1: Optional[nn.LayerNorm] = None, lin2: Optional[nn.Linear] = None) -> torch.Tensor:
    """lin2(relu(ln1(lin1(ln0(x))))) with the optional stages as given; torch modules when the fused
    kernels do not cover the widths (still on the GPU)."""
    assert (ln1 is None) == (lin2 is None)
    if not supported(x, lin1, lin2, (ln0, ln1)):
        h = wide_linear(x if ln0 is None else wide_layer_norm(x, ln0), lin1)
        return h if lin2 is None else wide_linear(wide_layer_norm(h, ln1, relu=True), lin2)
    eps = (ln0 or ln1).eps if (ln0 is not None or ln1 is not None) else 1e-5
    if ln0 is not None and ln1 is not None:
        assert ln0.eps == ln1.eps
    return _HeadFunction.apply(x, eps,
                               None if ln0 is None else ln0.weight, None if ln0 is None else ln0.bias,
                               lin1.weight, lin1.bias,
                               None if ln1 is None else ln1.weight, None if ln1 is None else ln1.bias,
                               None if lin2 is None else lin2.weight, None if lin2 is None else lin2.bias)


class _WideLinearFunction(torch.autograd.Function):
    """y = x W^T + b for heads wider than the fused kernels cover, every product on the library's fp32 MFMA
    GEMMs: forward (tgru_input_projection), dx = dy W (tgru_linear_input_grad), the weight gradient dy^T x -- an
    (out x in) product over B*T rows, split over the rows (tgru_linear_weight_grad) -- and the bias gradient as a
    column sum (tgru_column_sum)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.has_bias = b is not None
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = w.contiguous()
        ctx.save_for_backward(x2, w)
        ctx.xshape = x.shape
        if _TORCH_WIDE_GEMMS:
            return torch.nn.functional.linear(x, w, b)
        y = torch.empty(x.shape[:-1] + (w.shape[0],), device=x.device, dtype=torch.float32)
        with on_device(x.device):
            check(lib().tgru_input_projection(_ptr(x2), _ptr(w), _ptr(None if b is None else b.contiguous()), _ptr(y),
                                              x2.shape[0], w.shape[0], w.shape[1], _stream(x.device)),
                  "tgru_input_projection")
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        rows, out_f, in_f = dy2.shape[0], w.shape[0], w.shape[1]
        dev = x2.device
        dx = dw = db = None
        lb = lib()
        with on_device(dev):
            stream = _stream(dev)
            if ctx.needs_input_grad[0]:
                if _TORCH_WIDE_GEMMS:
                    dx = (dy2 @ w).view(ctx.xshape)
                else:
                    dx = torch.empty(ctx.xshape, device=dev, dtype=torch.float32)
                    check(lb.tgru_linear_input_grad(_ptr(dy2), _ptr(w), _ptr(dx), rows, out_f, in_f, stream),
                          "tgru_linear_input_grad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w, memory_format=torch.contiguous_format)
                ws = torch.empty(max(1, lb.tgru_linear_weight_grad_workspace_bytes(rows, out_f, in_f) // 4),
                                 device=dev, dtype=torch.float32)
                check(lb.tgru_linear_weight_grad(_ptr(dy2), _ptr(x2), _ptr(dw), _ptr(ws), rows, out_f, in_f, 0, stream),
                      "tgru_linear_weight_grad")
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if out_f % 4 == 0 and lb.tgru_layer_norm_supported(out_f):
                    db = torch.empty(out_f, device=dev, dtype=torch.float32)
                    ws = torch.empty(max(1, lb.tgru_column_sum_workspace_bytes(rows, out_f) // 4), device=dev,
                                     dtype=torch.float32)
                    check(lb.tgru_column_sum(_ptr(dy2), _ptr(db), _ptr(ws), rows, out_f, 0, stream), "tgru_column_sum")
                else:
                    db = dy2.sum(0)
        return dx, dw, db


class _WideLayerNormFunction(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (width a multiple of 4, at most 1024), optionally followed by ReLU in
    the same pass, on tgru_layer_norm_act_forward / _backward: each tensor crosses HBM once per pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty_like(x2)
        stats = torch.empty(2 * x2.shape[0], device=x.device, dtype=torch.float32)
        with on_device(x.device):
            check(lib().tgru_layer_norm_act_forward(_ptr(x2), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), x2.shape[0],
                                                    x2.shape[1], float(eps), int(relu), _stream(x.device)),
                  "tgru_layer_norm_act_forward")
        ctx.save_for_backward(x2, stats, gamma, beta)
        ctx.relu = int(relu)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, stats, gamma, beta = ctx.saved_tensors
        rows, width = x2.shape
        dy2 = dy.reshape(rows, width).contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dgamma = torch.empty_like(gamma) if (gamma is not None and ctx.needs_input_grad[1]) else None
        dbeta = torch.empty(width, device=x2.device, dtype=torch.float32) if (beta is not None and ctx.needs_input_grad[2]) else None
        lb = lib()
        ws = torch.empty(max(1, lb.tgru_layer_norm_backward_workspace_bytes(rows, width) // 4), device=x2.device,
                         dtype=torch.float32)
        with on_device(x2.device):
            check(lb.tgru_layer_norm_act_backward(_ptr(x2), _ptr(dy2), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(dx),
                                                  _ptr(dgamma), _ptr(dbeta), _ptr(ws), rows, width, ctx.relu, 0,
                                                  _stream(x2.device)), "tgru_layer_norm_act_backward")
        return (None if dx is None else dx.view(ctx.shape)), dgamma, dbeta, None, None


WIDE_LINEAR_MIN_ROWS = 2048
# A/B switch for measurements: True keeps torch's (vendor) GEMMs for y and dx of the heads beyond the fused
# kernels; the library still forms their weight / bias gradients and LayerNorms
_TORCH_WIDE_GEMMS = False


def wide_layer_norm(x: torch.Tensor, ln: nn.LayerNorm, relu: bool = False) -> torch.Tensor:
    """ln(x), or relu(ln(x)); on the GPU, from WIDE_LINEAR_MIN_ROWS rows on and for widths the library covers,
    through the library's LayerNorm kernels (one pass, the ReLU included)."""
    width = x.shape[-1]
    rows = x.numel() // max(1, width)
    if (x.is_cuda and x.dtype == torch.float32 and rows >= WIDE_LINEAR_MIN_ROWS and len(ln.normalized_shape) == 1
            and ln.normalized_shape[0] == width and lib().tgru_layer_norm_supported(width)
            and (ln.weight is None or ln.weight.dtype == torch.float32)):
        return _WideLayerNormFunction.apply(x, ln.weight, ln.bias, ln.eps, relu)
    return torch.relu(ln(x)) if relu else ln(x)


def wide_linear(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """lin(x); on the GPU, from WIDE_LINEAR_MIN_ROWS rows on, through the library's GEMMs.  A layer that asks for
    no gradient at all (no_grad, frozen parameters and a constant input) still takes the library's forward."""
    rows = x.numel() // max(1, x.shape[-1])
    if (x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32 and rows >= WIDE_LINEAR_MIN_ROWS
            and rows < (1 << 29)):
        return _WideLinearFunction.apply(x, lin.weight, lin.bias)
    return lin(x)


def _split_grads(dparams, din, dmid, dout, params):
    """Views of the flat gradient buffer in the order of `params` = (g0, b0, w1, c1, g1, b1, w2, c2)."""
    g0, b0, w1, c1, g1, b1, w2, c2 = params
    sizes = [("w1", dmid * din, (dmid, din), True), ("c1", dmid, (dmid,), True),
             ("g0", din, (din,), g0 is not None), ("b0", din, (din,), b0 is not None),
             ("g1", dmid, (dmid,), g1 is not None), ("b1", dmid, (dmid,), b1 is not None),
             ("w2", dout * dmid, (dout, dmid), w2 is not None), ("c2", dout, (dout,), c2 is not None)]
    out, o = {}, 0
    for name, cnt, shape, present in sizes:
        out[name] = dparams[o:o + cnt].view(shape) if present else None
        o += cnt
    return [out[k] for k in ("g0", "b0", "w1", "c1", "g1", "b1", "w2", "c2")]


class _HeadPairFunction(torch.autograd.Function):
    """(yA, yB) = two independent heads in one launch forward, one + one (reduce) backward.
    Arguments: epsA, epsB, then xA, g0, b0, w1, c1, g1, b1, w2, c2 of head A and the same nine of head B."""

    @staticmethod
    def forward(ctx, eps_a, eps_b, *tensors):
        heads, args = [], []
        for k, eps in enumerate((eps_a, eps_b)):
            x, *params = tensors[9 * k:9 * k + 9]
            g0, b0, w1, c1, g1, b1, w2, c2 = params
            din, dmid = x.shape[-1], w1.shape[0]
            dout = dmid if w2 is None else w2.shape[0]
            x2 = x.contiguous().view(-1, din)
            params = [None if p is None else p.contiguous() for p in params]
            y = torch.empty(x.shape[:-1] + (dout,), device=x.device, dtype=torch.float32)
            heads.append(dict(x2=x2, params=params, dims=(x2.shape[0], din, dmid, dout, float(eps)), y=y))
            args.append(HeadArgs(_addr(x2), _addr(y), None, None, x2.shape[0], din, dmid, dout,
                                 *[_addr(p) for p in params], float(eps), None, None, 0))
        dev = heads[0]["x2"].device
        with on_device(dev):
            check(lib().tgru_head_forward_pair(ctypes.byref(args[0]), ctypes.byref(args[1]), _stream(dev)),
                  "tgru_head_forward_pair")
        ctx.dims = [h["dims"] for h in heads]
        ctx.present = [[p is not None for p in h["params"]] for h in heads]
        saved = []
        for h in heads:
            saved += [h["x2"]] + [p for p in h["params"] if p is not None]
        ctx.save_for_backward(*saved)
        # autograd takes every output of a Function to depend on every input; a head none of whose own inputs
        # wants a gradient (a network that is not updated, fed by data) must not drag a backward pass behind it
        for k in range(2):
            if not any(ctx.needs_input_grad[2 + 9 * k:11 + 9 * k]):
                ctx.mark_non_differentiable(heads[k]["y"])
        return heads[0]["y"], heads[1]["y"]

    @staticmethod
    def backward(ctx, dy_a, dy_b):
        saved = list(ctx.saved_tensors)
        dev = saved[0].device
        args, results, keep, live = [], [], [], []
        queue = deferred.active()
        with on_device(dev):
            lb = lib()
            heads = []
            # per head: does any parameter want a gradient, does anything at all (a head that owes nothing is
            # left out: a network the step does not update, fed by data)
            want_p = [any(ctx.needs_input_grad[3 + 9 * k:11 + 9 * k]) for k in range(2)]
            owes = [want_p[k] or ctx.needs_input_grad[2 + 9 * k] for k in range(2)]
            for k in range(2):
                x2 = saved.pop(0)
                params = [saved.pop(0) if pres else None for pres in ctx.present[k]]
                heads.append((x2, params))
                if want_p[k]:
                    live += [p for p in params if p is not None]
            defer = bool(live) and queue is not None and queue.may_defer(live)
            flag = _lib.DEFER_REDUCE if defer else 0
            for k, dy in enumerate((dy_a, dy_b)):
                if not owes[k]:
                    results.append((None, [None] * 8))
                    continue
                M, din, dmid, dout, eps = ctx.dims[k]
                x2, params = heads[k]
                dy2 = dy.contiguous().view(-1, dout)
                dparams = None
                if want_p[k]:
                    nfl = lb.tgru_head_grad_floats(din, dmid, dout)
                    dparams = queue.grad_buffer(nfl, dev) if defer else torch.empty(nfl, device=dev, dtype=torch.float32)
                ws = torch.empty(lb.tgru_head_backward_workspace_bytes(M, din, dmid, dout) // 4, device=dev,
                                 dtype=torch.float32)
                dx = torch.empty_like(x2) if ctx.needs_input_grad[2 + 9 * k] else None
                keep += [dy2, ws, dparams]
                args.append(HeadArgs(_addr(x2), None, _addr(dy2), _addr(dx), M, din, dmid, dout,
                                     *[_addr(p) for p in params], eps, _addr(dparams), _addr(ws), flag))
                grads = _split_grads(dparams, din, dmid, dout, params) if want_p[k] else [None] * 8
                results.append((None if dx is None else dx.view(dy.shape[:-1] + (din,)), grads))
            if len(args) == 2:
                check(lb.tgru_head_backward_pair(ctypes.byref(args[0]), ctypes.byref(args[1]), _stream(dev)),
                      "tgru_head_backward_pair")
            else:
                a = args[0]
                check(lb.tgru_head_backward(a.x, a.dy, a.dx, a.M, a.din, a.dmid, a.dout, a.ln0_w, a.ln0_b, a.w1, a.b1,
                                            a.ln1_w, a.ln1_b, a.w2, a.b2, a.eps, a.dparams, a.workspace,
                                            a.accumulate, _stream(dev)), "tgru_head_backward")
            if defer:
                for a in args:
                    queue.add(lb.tgru_head_reduce_jobs, a, keep, live, dev)
        out = [None, None]
        for dx, grads in results:
            out += [dx, *grads]
        return tuple(out)


def _flat_params(ln0, lin1, ln1, lin2):
    return (None if ln0 is None else ln0.weight, None if ln0 is None else ln0.bias, lin1.weight, lin1.bias,
            None if ln1 is None else ln1.weight, None if ln1 is None else ln1.bias,
            None if lin2 is None else lin2.weight, None if lin2 is None else lin2.bias)


def _eps_of(ln0, ln1):
    if ln0 is not None and ln1 is not None:
        assert ln0.eps == ln1.eps
    return (ln0 or ln1).eps if (ln0 is not None or ln1 is not None) else 1e-5


def head_apply_pair(x_a: torch.Tensor, spec_a, x_b: torch.Tensor, spec_b):
    """(head_apply(x_a, *spec_a), head_apply(x_b, *spec_b)) for two independent heads, in one launch when
    the fused kernels cover both; spec = (ln0, lin1, ln1, lin2) as for head_apply."""
    ok = all(supported(x, spec[1], spec[3], (spec[0], spec[2])) for x, spec in ((x_a, spec_a), (x_b, spec_b)))
    if not ok or x_a.device != x_b.device:
        return head_apply(x_a, *spec_a), head_apply(x_b, *spec_b)
    for spec in (spec_a, spec_b):
        assert (spec[2] is None) == (spec[3] is None)
    return _HeadPairFunction.apply(_eps_of(spec_a[0], spec_a[2]), _eps_of(spec_b[0], spec_b[2]),
                                   x_a, *_flat_params(*spec_a), x_b, *_flat_params(*spec_b))
