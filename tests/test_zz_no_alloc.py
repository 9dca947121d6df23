"""tgru_forward_static -- tgru_forward without a workspace argument, for callers that cannot allocate --
against tgru_forward itself, through the C ABI: same kernels, so the outputs must be bit-identical.
Two calls in a row share the library's scratch array; the second is smaller and must not see the first."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ptr_array(ts):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def _both(B, T, In, H, L, layout, seed, with_reserve, lengths=None):
    """(y, h_n, reserve) from tgru_forward with a caller workspace and from tgru_forward_static."""
    import timegan_amd as tg
    from bench import make_weights
    lb = tg.lib()
    dev = torch.device("cuda:0")
    w = make_weights(In, H, L, dev, seed)
    arrays = [_ptr_array(w[i::4]) for i in range(4)]
    g = torch.Generator().manual_seed(seed)
    shape = (T, B) if layout == tg.LAYOUT_TBF else (B, T)
    x = torch.rand(shape + (In,), generator=g).to(dev)
    lens = None if lengths is None else torch.tensor(lengths, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    outs = []
    for static in (False, True):
        y = torch.full(shape + (H,), float("nan"), device=dev)
        h_n = torch.full((L, B, H), float("nan"), device=dev)
        n_res = lb.tgru_reserve_bytes(B, T, In, H, L, layout) // 4
        reserve = torch.full((n_res,), float("nan"), device=dev) if with_reserve else None
        if static:
            rc = lb.tgru_forward_static(_ptr(x), None, _ptr(lens), None, *arrays, _ptr(y), _ptr(h_n), _ptr(reserve),
                                        B, T, In, H, L, layout, stream)
        else:
            ws = torch.full((lb.tgru_forward_workspace_bytes(B, T, In, H, L) // 4,), float("nan"), device=dev)
            rc = lb.tgru_forward(_ptr(x), None, _ptr(lens), None, *arrays, _ptr(y), _ptr(h_n), _ptr(reserve),
                                 _ptr(ws), B, T, In, H, L, layout, stream)
        assert rc == 0, lb.tgru_error_string(rc)
        torch.cuda.synchronize()
        outs.append((y, h_n, reserve))
    return outs


def _same(a, b, what):
    # bit-identical, NaN-aware (unwritten parts of `reserve` keep the NaN fill in both)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), what


def test_static_forward_equals_forward_on_two_calls_in_a_row():
    import timegan_amd as tg
    # first the energy stack at its real batch (47 MB of scratch: input projections, ping-pong layer outputs),
    # then a smaller call that reuses the front of the same array: mid-size kernels, then the per-step kernels
    cases = [(1024, 24, 96, 96, 3, tg.LAYOUT_TBF, 1, False, None),
             (37, 9, 28, 96, 2, tg.LAYOUT_BTF, 2, True, None),
             (130, 5, 40, 160, 2, tg.LAYOUT_BTF, 3, True, [5] * 60 + [3] * 60 + [1] * 10)]
    for (B, T, In, H, L, layout, seed, with_reserve, lengths) in cases:
        (y0, h0, r0), (y1, h1, r1) = _both(B, T, In, H, L, layout, seed, with_reserve, lengths)
        tag = f"B{B} T{T} In{In} H{H} L{L}"
        assert not torch.isnan(y1).any(), tag + ": y has unwritten elements"
        _same(y1, y0, tag + " y")
        _same(h1, h0, tag + " h_n")
        if with_reserve:
            _same(r1, r0, tag + " reserve")


def test_static_forward_rejects_what_does_not_fit_and_launches_nothing():
    import timegan_amd as tg
    from timegan_amd import _lib
    lb = tg.lib()
    dev = torch.device("cuda:0")
    B, T, In, H, L = 2048, 256, 512, 512, 3          # the stress shard: 6.4 GB of input projections
    assert lb.tgru_forward_workspace_bytes(B, T, In, H, L) > _lib.STATIC_WORKSPACE_BYTES
    fake = torch.zeros(64, device=dev)
    arr = _ptr_array([fake] * L)
    n0 = tg.launch_count()
    rc = lb.tgru_forward_static(_ptr(fake), None, None, None, arr, arr, arr, arr, _ptr(fake), None, None,
                                B, T, In, H, L, tg.LAYOUT_TBF, None)
    assert rc == -7 and b"TGRU_STATIC_WORKSPACE_BYTES" in lb.tgru_error_string(rc)
    assert tg.launch_count() == n0
