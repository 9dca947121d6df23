"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WNAMES = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")

# Tolerances of the parity bar: forward rtol 1e-5, gradients rtol 1e-4, each with an absolute
# floor of 1e-6 (forward) / 1e-5 (gradients) times the largest reference magnitude, because fp32
# results near zero carry rounding noise of that size (fp32 vs fp64 CPU torch differ by ~6e-7
# relative to max|ref| on these shapes).
FWD_RTOL, FWD_ATOL_SCALE = 1e-5, 1e-6
GRAD_RTOL, GRAD_ATOL_SCALE = 1e-4, 1e-5


def gru_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "gru_*.npz")))


def load_gru_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, T, In, H, L = (int(v) for v in d["dims"])
    weights = [tuple(d[f"{n}_l{l}"] for n in WNAMES) for l in range(L)]
    return {
        "dims": (B, T, In, H, L), "x": d["x"], "dy": d["dy"], "weights": weights,
        "lengths": d["lengths"] if "lengths" in d else None,
        "dhn": d["dhn"] if "dhn" in d else None,
        "raw": d,
    }


def expected(case, tag):
    d = case["raw"]
    L = case["dims"][4]
    out = {"y": d[f"{tag}_y"], "h_n": d[f"{tag}_h_n"], "dx": d[f"{tag}_dx"]}
    out["grads"] = [tuple(d[f"{tag}_d{n}_l{l}"] for n in WNAMES) for l in range(L)]
    return out


def assert_close(got, ref, rtol, atol_scale, what, atol_abs=0.0):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    atol = atol_scale * max(float(np.abs(ref).max()), 1e-30) + atol_abs
    err = np.abs(got - ref)
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.size} elements out of tolerance, "
                           f"max abs err {err.max():.3e}, max |ref| {np.abs(ref).max():.3e}")
