"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/gru_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does; it fails loudly when its HIP library is missing.

The oracle restates the multi-layer GRU forward pass and BPTT that the reference runs through
torch.nn.GRU (outdated_files/TimeGANv1.py:35-38,61 and the four clones of that call site).
It is pinned against torch 2.10.0 CPU nn.GRU outputs committed under tests/golden.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgru_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a checker build, not a product build)."""
    src = [os.path.join(_HERE, f) for f in ("gru_oracle.c", "gru_oracle_impl.h", "philox_oracle.c")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        for name in ("gru_oracle_run_f32", "gru_oracle_run_f64"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5
        lib.philox4x32_10.restype = None
        lib.philox4x32_10.argtypes = [ctypes.c_void_p] * 3
        lib.philox_uniform_f32.restype = None
        lib.philox_uniform_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        lib.gru_oracle_param_count.restype = ctypes.c_int64
        lib.gru_oracle_param_count.argtypes = [ctypes.c_int] * 3
        _lib = lib
    return _lib


def flatten_params(weights, dtype):
    """weights: list over layers of (w_ih, w_hh, b_ih, b_hh) arrays -> flat oracle layout."""
    return np.concatenate([np.asarray(a, dtype=dtype).ravel() for layer in weights for a in layer])


def unflatten_params(flat, In, H, L):
    out, off = [], 0
    for l in range(L):
        in_l = In if l == 0 else H
        shapes = [(3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)]
        layer = []
        for s in shapes:
            n = int(np.prod(s))
            layer.append(flat[off:off + n].reshape(s).copy())
            off += n
        out.append(tuple(layer))
    return out


_ERRORS = {
    -1: "bad dimensions",
    -2: "Length of all samples has to be greater than 0, but found an element in 'lengths' that is <= 0",
    -3: "'lengths' array must be sorted in decreasing order when 'enforce_sorted' is True",
    -4: "out of memory",
}


def gru(x, weights, lengths=None, dy=None, dhn=None, dtype=np.float32):
    """Run the oracle.  x (B,T,In) batch-major; weights as for flatten_params.

    Returns dict(y (B,T,H), h_n (L,B,H)) and, when dy is given, dx plus `grads`
    (list over layers of (dw_ih, dw_hh, db_ih, db_hh)).
    """
    lib = _load()
    dtype = np.dtype(dtype)
    x = np.ascontiguousarray(x, dtype=dtype)
    B, T, In = x.shape
    L = len(weights)
    H = np.asarray(weights[0][1]).shape[1]
    params = flatten_params(weights, dtype)
    assert params.size == lib.gru_oracle_param_count(In, H, L)
    y = np.zeros((B, T, H), dtype=dtype)
    h_n = np.zeros((L, B, H), dtype=dtype)
    len_arr = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.int64)

    def ptr(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    dx = grads = None
    if dy is not None:
        dy = np.ascontiguousarray(dy, dtype=dtype)
        dhn = None if dhn is None else np.ascontiguousarray(dhn, dtype=dtype)
        dx = np.zeros_like(x)
        grads = np.zeros_like(params)
    fn = lib.gru_oracle_run_f32 if dtype == np.float32 else lib.gru_oracle_run_f64
    rc = fn(ptr(x), ptr(len_arr), ptr(params), ptr(y), ptr(h_n),
            ptr(dy), ptr(dhn), ptr(dx), ptr(grads), B, T, In, H, L)
    if rc != 0:
        raise RuntimeError(_ERRORS.get(rc, f"oracle error {rc}"))
    out = {"y": y, "h_n": h_n}
    if dy is not None:
        out["dx"] = dx
        out["grads"] = unflatten_params(grads, In, H, L)
    return out


def philox4x32_10(ctr, key):
    """One Philox4x32-10 block: four 32-bit counter words, two key words -> four output words."""
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    _load().philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def uniform(n, seed, stream, first=0):
    """Elements [first, first + n) of noise stream `stream` under `seed`, as tgru_uniform defines them."""
    out = np.empty(int(n), dtype=np.float32)
    _load().philox_uniform_f32(out.ctypes.data, int(n), int(seed), int(stream), int(first))
    return out
