"""ctypes binding of libtgru.so, the C-ABI HIP library declared in include/tgru.h.

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TGRU_LIB_PATH selects another build of the library (A/B measurements); the default is the in-tree one
LIB_PATH = os.environ.get("TGRU_LIB_PATH") or os.path.join(_HERE, "libtgru.so")

LAYOUT_TBF = 0
LAYOUT_BTF = 1
LAYOUT_PACKED = 2

_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int



class ForwardArgs(ctypes.Structure):
    """tgru_forward_args of include/tgru.h."""
    _fields_ = [(n, _vp) for n in ("x", "h0", "lengths", "toff", "w_ih", "w_hh", "b_ih", "b_hh",
                                   "y", "h_n", "reserve", "workspace")] + \
               [(n, _i) for n in ("B", "T", "In", "H", "L", "layout")]


class BackwardArgs(ctypes.Structure):
    """tgru_backward_args of include/tgru.h."""
    _fields_ = [(n, _vp) for n in ("dy", "dh_n", "x", "h0", "y", "lengths", "toff", "w_ih", "w_hh", "b_ih", "b_hh",
                                   "reserve", "workspace", "dx", "dh0", "dw_ih", "dw_hh", "db_ih", "db_hh")] + \
               [(n, _i) for n in ("B", "T", "In", "H", "L", "layout", "accumulate")]


class HeadArgs(ctypes.Structure):
    """tgru_head_args of include/tgru.h."""
    _fields_ = [("x", _vp), ("y", _vp), ("dy", _vp), ("dx", _vp)] + \
               [(n, _i) for n in ("M", "din", "dmid", "dout")] + \
               [(n, _vp) for n in ("ln0_w", "ln0_b", "w1", "b1", "ln1_w", "ln1_b", "w2", "b2")] + \
               [("eps", ctypes.c_float), ("dparams", _vp), ("workspace", _vp), ("accumulate", _i)]


class ReduceJob(ctypes.Structure):
    """tgru_reduce_job of include/tgru.h."""
    _fields_ = [("src", _vp), ("stride", ctypes.c_int64), ("count", ctypes.c_int32), ("n", ctypes.c_int32),
                ("dst", _vp)]


class AdamJob(ctypes.Structure):
    """tgru_adam_job of include/tgru.h."""
    _fields_ = [(n, _vp) for n in ("param", "grad", "exp_avg", "exp_avg_sq", "step")] + \
               [("n", ctypes.c_int32), ("reserved", ctypes.c_int32)]


MAX_ADAM_JOBS = 64        # TGRU_MAX_ADAM_JOBS
MAX_ADAM_ELEMS = 65536    # TGRU_MAX_ADAM_ELEMS
DEFER_REDUCE = 2          # TGRU_DEFER_REDUCE
STATIC_WORKSPACE_BYTES = 64 << 20   # TGRU_STATIC_WORKSPACE_BYTES
MAX_JOBS_PER_CALL = 32    # upper bound of jobs one backward call can owe (5 per layer, at most 4 layers on that path)

# name -> (restype, argtypes); mirrors include/tgru.h one to one
SIGNATURES = {
    "tgru_version": (_i, []),
    "tgru_launch_count": (ctypes.c_ulonglong, []),
    "tgru_reload_switches": (None, []),
    "tgru_persist_timeouts": (ctypes.c_long, [_i]),
    "tgru_error_string": (ctypes.c_char_p, [_i]),
    "tgru_reserve_bytes": (ctypes.c_size_t, [_i] * 6),
    "tgru_forward_workspace_bytes": (ctypes.c_size_t, [_i] * 5),
    "tgru_backward_workspace_bytes": (ctypes.c_size_t, [_i] * 5),
    "tgru_forward": (_i, [_vp] * 12 + [_i] * 6 + [_vp]),
    "tgru_forward_static": (_i, [_vp] * 11 + [_i] * 6 + [_vp]),
    "tgru_backward": (_i, [_vp] * 19 + [_i] * 7 + [_vp]),
    "tgru_forward_pair": (_i, [ctypes.POINTER(ForwardArgs)] * 2 + [_vp]),
    "tgru_backward_pair": (_i, [ctypes.POINTER(BackwardArgs)] * 2 + [_vp]),
    "tgru_input_projection": (_i, [_vp] * 4 + [_i] * 3 + [_vp]),
    "tgru_layer_norm_supported": (_i, [_i]),
    "tgru_layer_norm_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long, _i]),
    "tgru_layer_norm_forward": (_i, [_vp] * 5 + [ctypes.c_long, _i, ctypes.c_float, _vp]),
    "tgru_layer_norm_backward": (_i, [_vp] * 8 + [ctypes.c_long, _i, _i, _vp]),
    "tgru_layer_norm_act_forward": (_i, [_vp] * 5 + [ctypes.c_long, _i, ctypes.c_float, _i, _vp]),
    "tgru_layer_norm_act_backward": (_i, [_vp] * 9 + [ctypes.c_long, _i, _i, _i, _vp]),
    "tgru_column_sum_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long, _i]),
    "tgru_column_sum": (_i, [_vp] * 3 + [ctypes.c_long, _i, _i, _vp]),
    "tgru_linear_input_grad": (_i, [_vp] * 3 + [_i] * 3 + [_vp]),
    "tgru_linear_weight_grad_workspace_bytes": (ctypes.c_size_t, [_i] * 3),
    "tgru_linear_weight_grad": (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    "tgru_head_supported": (_i, [_i] * 4),
    "tgru_head_grad_floats": (ctypes.c_size_t, [_i] * 3),
    "tgru_head_backward_workspace_bytes": (ctypes.c_size_t, [_i] * 4),
    "tgru_head_forward": (_i, [_vp] * 2 + [_i] * 4 + [_vp] * 8 + [ctypes.c_float, _vp]),
    "tgru_head_backward": (_i, [_vp] * 3 + [_i] * 4 + [_vp] * 8 + [ctypes.c_float, _vp, _vp, _i, _vp]),
    "tgru_head_forward_pair": (_i, [ctypes.POINTER(HeadArgs)] * 2 + [_vp]),
    "tgru_head_backward_pair": (_i, [ctypes.POINTER(HeadArgs)] * 2 + [_vp]),
    "tgru_backward_reduce_jobs": (_i, [ctypes.POINTER(BackwardArgs), ctypes.POINTER(ReduceJob), _i]),
    "tgru_head_reduce_jobs": (_i, [ctypes.POINTER(HeadArgs), ctypes.POINTER(ReduceJob), _i]),
    "tgru_reduce_jobs": (_i, [ctypes.POINTER(ReduceJob), _i, _vp]),
    "tgru_adam_jobs": (_i, [ctypes.POINTER(AdamJob), _i] + [ctypes.c_float] * 4 + [_vp, _vp]),
    "tgru_uniform": (_i, [_vp, ctypes.c_long, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp]),
    "tgru_loss_layout": (_i, [_i] * 3 + [ctypes.POINTER(ctypes.c_int64)]),
    "tgru_loss_partial": (_i, [_i, _i, _vp] + [_i] * 4 + [_vp, _vp]),
    "tgru_loss_final": (_i, [_i] * 5 + [ctypes.c_float, _vp, _vp]),
    "tgru_loss_backward": (_i, [_i, _vp, _vp, _vp] + [_i] * 4 + [ctypes.c_float, _vp, _vp]),
}


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """Context in which `device` is the current GPU.  The library launches on the current device; when it
    already is `device` (the usual case) nothing is switched -- torch.cuda.device() costs ~10 us a use."""
    import torch
    if torch.cuda.current_device() == device.index:
        return _NO_SWITCH
    return torch.cuda.device(device)


class TgruError(RuntimeError):
    pass


def lib():
    """Load libtgru.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgruError(
                f"{LIB_PATH} not found: the HIP library is not built "
                "(run `make -C time-gan_amd/csrc`); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def reload_switches():
    """Make the library read its TGRU_* environment switches again (it reads them once per process)."""
    lib().tgru_reload_switches()


def launch_count() -> int:
    """Kernel launches the library has issued in this process so far."""
    return int(lib().tgru_launch_count())


def persist_timeouts(device=None, clear: bool = True) -> int:
    """Calls on `device` (default: the current one) whose persistent H > 128 kernels gave up waiting for each
    other since the count was last cleared; their results were overwritten with NaN (tgru_persist_timeouts of
    include/tgru.h).  Synchronises the device first, so every call issued so far is counted."""
    import torch
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with on_device(device):
        torch.cuda.synchronize(device)
        n = int(lib().tgru_persist_timeouts(1 if clear else 0))
    if n < 0:
        raise TgruError("tgru_persist_timeouts failed (no gfx950 device, or the device copy failed)")
    return n


def check_persist_timeouts(device=None):
    """Raises TgruError if a persistent-kernel wait has timed out on `device` since the last check."""
    n = persist_timeouts(device, clear=True)
    if n:
        raise TgruError(
            f"{n} tgru call(s) on this GPU ended with a timed-out wait of the persistent H > 128 kernels: their "
            "workgroups were not all resident (another kernel or process held compute units); the results of "
            "those calls were set to NaN.  TGRU_NO_PERSIST=1 selects kernels without that requirement")


def check(code: int, what: str):
    if code != 0:
        msg = lib().tgru_error_string(code).decode()
        raise TgruError(f"{what} failed: {msg} (code {code})")
