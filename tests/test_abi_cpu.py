"""CPU tests of the boundary: the C-ABI library loads and exports every symbol the header declares,
its size queries behave, and the host-side argument validation mirrors torch's errors.
No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tgru.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgru_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import timegan_amd as tg
    from timegan_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/tgru.h"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/tgru.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "python binding table and header disagree"
    assert tg.lib().tgru_version() >= 100


def test_size_queries_and_error_strings():
    import timegan_amd as tg
    lb = tg.lib()
    B, T, In, H, L = 128, 24, 24, 24, 3
    tbh = B * T * H * 4
    # 4 gate planes per layer + (L-1) intermediate outputs (+1 copy when the layout is not time-major)
    assert lb.tgru_reserve_bytes(B, T, In, H, L, tg.LAYOUT_TBF) >= (4 * L + L - 1) * tbh
    assert lb.tgru_reserve_bytes(B, T, In, H, L, tg.LAYOUT_BTF) >= (4 * L + L) * tbh
    assert lb.tgru_forward_workspace_bytes(B, T, In, H, L) > 0
    assert lb.tgru_backward_workspace_bytes(B, T, In, H, L) > 0
    assert lb.tgru_reserve_bytes(0, T, In, H, L, 0) == 0            # bad dims -> 0
    assert lb.tgru_reserve_bytes(B, T, In, 4096, L, 0) == 0         # hidden size beyond the built range
    assert lb.tgru_error_string(0) == b"ok"
    for code in range(-7, 0):
        assert lb.tgru_error_string(code) not in (b"ok", b"unknown error")
    # NULL pointers are rejected before anything is launched
    rc = lb.tgru_forward(None, None, None, None, None, None, None, None, None, None, None, None,
                         B, T, In, H, L, 0, None)
    assert rc == -2


def test_cpu_tensors_fail_loudly_no_fallback():
    import timegan_amd as tg
    gru = tg.GRU(4, 8, 1, batch_first=True)
    with pytest.raises(tg.TgruError, match="no CPU fallback"):
        gru(torch.zeros(2, 3, 4))


def test_lengths_validation_matches_torch_messages():
    from timegan_amd.functional import lengths_to_device
    assert lengths_to_device([5, 5, 5], 3, 5, "cpu") is None
    assert lengths_to_device(torch.ones(3) * 5, 3, 5, "cpu") is None       # float tensor accepted
    out = lengths_to_device([5, 3, 1], 3, 5, "cpu")
    assert out.dtype == torch.int32 and out.tolist() == [5, 3, 1]
    with pytest.raises(RuntimeError, match="sorted in decreasing order"):
        lengths_to_device([1, 3, 5], 3, 5, "cpu")
    with pytest.raises(RuntimeError, match="greater than 0"):
        lengths_to_device([3, 1, 0], 3, 5, "cpu")


def test_gru_module_matches_torch_parameter_layout_and_init():
    import timegan_amd as tg
    torch.manual_seed(3)
    ours = tg.GRU(6, 24, 3, batch_first=True)
    torch.manual_seed(3)
    ref = torch.nn.GRU(6, 24, 3, batch_first=True)
    sd_o, sd_r = ours.state_dict(), ref.state_dict()
    assert list(sd_o) == list(sd_r)
    for k in sd_r:
        assert sd_o[k].shape == sd_r[k].shape
        assert torch.equal(sd_o[k], sd_r[k]), f"{k}: same seed must give the same initial values"
    ref.load_state_dict(sd_o)       # interchange both ways
    ours.load_state_dict(sd_r)


def test_linear_weight_grad_argument_checks():
    from timegan_amd import _lib
    lb = _lib.lib()
    assert lb.tgru_linear_weight_grad_workspace_bytes(0, 96, 96) == 0
    assert lb.tgru_linear_weight_grad_workspace_bytes(24576, 96, 96) >= 96 * 96 * 4
    assert lb.tgru_linear_weight_grad(None, None, None, None, 16, 8, 8, 0, None) != 0      # NULL pointers
    assert lb.tgru_linear_weight_grad(None, None, None, None, 0, 8, 8, 0, None) != 0       # no rows


def test_argument_checks_of_the_struct_entry_points():
    """The pair / head / loss / reduce entry points reject bad arguments before anything is launched."""
    import timegan_amd as tg
    from timegan_amd._lib import BackwardArgs, ForwardArgs, HeadArgs, ReduceJob
    lb = tg.lib()
    E_DIMS, E_NULL = -1, -2
    fa, ba, ha = ForwardArgs(), BackwardArgs(), HeadArgs()          # all-zero structs
    assert lb.tgru_forward_pair(None, None, None) == E_NULL
    assert lb.tgru_forward_pair(ctypes.byref(fa), ctypes.byref(fa), None) == E_DIMS
    assert lb.tgru_backward_pair(ctypes.byref(ba), ctypes.byref(ba), None) == E_DIMS
    fa.B, fa.T, fa.In, fa.H, fa.L = 4, 3, 5, 6, 1
    assert lb.tgru_forward_pair(ctypes.byref(fa), ctypes.byref(fa), None) == E_NULL
    assert lb.tgru_head_forward_pair(None, ctypes.byref(ha), None) == E_NULL
    assert lb.tgru_head_forward_pair(ctypes.byref(ha), ctypes.byref(ha), None) == E_DIMS
    ha.M, ha.din, ha.dmid, ha.dout = 8, 129, 8, 8                   # input wider than the kernels cover
    assert lb.tgru_head_forward_pair(ctypes.byref(ha), ctypes.byref(ha), None) == E_DIMS
    ha.din, ha.dmid = 8, 193                                        # middle wider than the kernels cover
    assert lb.tgru_head_backward_pair(ctypes.byref(ha), ctypes.byref(ha), None) == E_DIMS
    ha.din, ha.dmid = 33, 8                                         # covered since the LDS-resident kernels
    assert lb.tgru_head_forward_pair(ctypes.byref(ha), ctypes.byref(ha), None) == E_NULL
    assert lb.tgru_head_supported(24, 48, 24, 1) == 1 and lb.tgru_head_supported(96, 96, 28, 1) == 1
    assert lb.tgru_head_supported(28, 192, 96, 1) == 1 and lb.tgru_head_supported(96, 1, 1, 0) == 1
    assert lb.tgru_head_supported(96, 96, 28, 0) == 0               # no second Linear: dout must equal dmid
    assert lb.tgru_head_supported(128, 192, 128, 1) == 0            # the padded pair of matrices exceeds the LDS budget
    assert lb.tgru_head_backward_workspace_bytes(1024, 96, 96, 96) >= 3 * 1024 * 96 * 4
    assert lb.tgru_head_grad_floats(24, 24, 6) == 24 * 24 + 24 + 24 + 24 + 24 + 24 + 6 * 24 + 6
    jobs = (ReduceJob * 2)()
    assert lb.tgru_reduce_jobs(jobs, 0, None) == 0                  # nothing to do
    assert lb.tgru_reduce_jobs(None, 2, None) == E_NULL
    assert lb.tgru_reduce_jobs(jobs, 2, None) == E_NULL             # jobs without source / destination
    assert lb.tgru_backward_reduce_jobs(None, jobs, 2) == E_NULL
    lay = (ctypes.c_int64 * 4)()
    assert lb.tgru_loss_layout(4, 24, 6, lay) == 0                  # generator loss: 2 x 144 column sums + squares
    n_a, n_b, off_out, total = (int(v) for v in lay)
    assert n_b == 2 * 24 * 6 and n_a == 4 + 3 * 256 + n_b and off_out == n_a + n_b and total > off_out + 3
    assert lb.tgru_loss_layout(0, 24, 6, lay) == 0 and int(lay[1]) == 0
    assert lb.tgru_loss_layout(7, 24, 6, lay) == E_DIMS
    assert lb.tgru_loss_final(0, 4, 3, 5, 6, 1.0, None, None) == E_NULL


def test_no_gfx950_device_is_reported_not_launched():
    """With valid-looking arguments but no gfx950 GPU as the current device, tgru_forward / tgru_backward
    return TGRU_E_NOGPU before anything is launched (the pointers below are never dereferenced)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the device check passes here")
    import timegan_amd as tg
    lb = tg.lib()
    E_NOGPU = -6
    B, T, In, H, L = 4, 3, 8, 8, 1
    fake = ctypes.c_void_p(0x1000)            # 16-byte aligned, non-NULL
    arr = (ctypes.c_void_p * L)(0x1000)
    rc = lb.tgru_forward(fake, None, None, None, arr, arr, arr, arr, fake, None, fake, fake,
                         B, T, In, H, L, tg.LAYOUT_TBF, None)
    assert rc == E_NOGPU
    assert b"gfx950" in lb.tgru_error_string(rc)
    rc = lb.tgru_backward(fake, None, fake, None, fake, None, None, arr, arr, arr, arr, fake, fake, fake, None,
                          arr, arr, arr, arr, B, T, In, H, L, tg.LAYOUT_TBF, 0, None)
    assert rc == E_NOGPU
    n0 = tg.launch_count()
    assert tg.launch_count() == n0            # nothing was launched by the rejected calls


def test_cpp_autograd_binding_loads_and_refuses_cpu_tensors():
    """time-gan_amd/_tgru_torch.so (csrc_torch/tgru_torch.cpp) is the torch extension over the same C ABI: it must
    load where the library loads, and like the ctypes binding it has no CPU path."""
    import torch
    from timegan_amd import functional
    mod = functional.cpp_op()
    assert mod is not None, "time-gan_amd/_tgru_torch.so is not built (python __graft_entry__.py build)"
    H = 8
    w = [torch.zeros(3 * H, 4), torch.zeros(3 * H, H), torch.zeros(3 * H), torch.zeros(3 * H)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mod.gru_apply(torch.zeros(2, 3, 4), None, None, None, 2, 3, 1, w)
