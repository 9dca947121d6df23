"""GPU parity tests: the HIP path, called through the C ABI, against the golden fixtures
(torch 2.10 CPU nn.GRU used as the reference does) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import (FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close, expected,
                     gru_case_names, load_gru_case)

pytestmark = pytest.mark.gpu

LAYOUTS = {"tbf": 0, "btf": 1, "packed": 2}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _flat_weights(weights, dev):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for layer in weights for a in layer]


def _to_layout(a_btf, layout, lengths):
    """(B,T,F) numpy -> tensor rows in `layout` (+ lengths/toff numpy for packed)."""
    B, T, F = a_btf.shape
    if layout == "btf":
        return a_btf
    if layout == "tbf":
        return np.ascontiguousarray(a_btf.transpose(1, 0, 2))
    lens = [T] * B if lengths is None else list(lengths)
    rows = [a_btf[b, t] for t in range(T) for b in range(B) if t < lens[b]]
    return np.stack(rows)


def _from_layout(a, layout, B, T, lengths):
    """inverse of _to_layout; padded positions zero."""
    F = a.shape[-1]
    if layout == "btf":
        return a
    if layout == "tbf":
        return a.transpose(1, 0, 2)
    lens = [T] * B if lengths is None else list(lengths)
    out = np.zeros((B, T, F), dtype=a.dtype)
    i = 0
    for t in range(T):
        for b in range(B):
            if t < lens[b]:
                out[b, t] = a[i]
                i += 1
    return out


def _run(case, layout):
    import timegan_amd as tg
    dev = _dev()
    B, T, In, H, L = case["dims"]
    lengths = case["lengths"]
    x = torch.from_numpy(_to_layout(case["x"], layout, lengths)).to(dev)
    dy_np = case["dy"].copy()
    if lengths is not None:
        for b, n in enumerate(lengths):
            dy_np[b, n:] = np.nan if layout != "packed" else 0   # padded-row gradients must be ignored
    dy = torch.from_numpy(_to_layout(dy_np, layout, lengths)).to(dev)
    w = _flat_weights(case["weights"], dev)
    dhn = None if case["dhn"] is None else torch.from_numpy(case["dhn"]).to(dev)
    lens_dev = toff_dev = None
    if lengths is not None or layout == "packed":
        lens = [T] * B if lengths is None else [int(v) for v in lengths]
        lens_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
        bs = [sum(1 for n in lens if n > t) for t in range(T)]
        toff_dev = torch.tensor(np.concatenate([[0], np.cumsum(bs)]), dtype=torch.int32, device=dev)
    out = tg.gru_forward_backward_raw(x, w, B, T, LAYOUTS[layout], dy, dh_n=dhn, lengths=lens_dev,
                                      toff=toff_dev)
    torch.cuda.synchronize()
    res = {"y": _from_layout(out["y"].cpu().numpy(), layout, B, T, lengths),
           "h_n": out["h_n"].cpu().numpy(),
           "dx": _from_layout(out["dx"].cpu().numpy(), layout, B, T, lengths),
           "grads": [tuple(g.cpu().numpy() for g in out["grads"][4 * l:4 * l + 4]) for l in range(L)]}
    return res


def _compare(res, ref, tag):
    assert_close(res["y"], ref["y"], FWD_RTOL, FWD_ATOL_SCALE, f"{tag} y")
    assert_close(res["h_n"], ref["h_n"], FWD_RTOL, FWD_ATOL_SCALE, f"{tag} h_n")
    assert_close(res["dx"], ref["dx"], GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} dx")
    for l, (g, r) in enumerate(zip(res["grads"], ref["grads"])):
        for name, a, b in zip(("dweight_ih", "dweight_hh", "dbias_ih", "dbias_hh"), g, r):
            assert_close(a, b, GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} {name}_l{l}")


@pytest.fixture(params=["auto", "h24", "small", "per_layer", "per_layer_guarded"])
def kernel_path(request, monkeypatch):
    """auto: the kernels the library picks for the shape (H, In <= 32: one sequence per workgroup,
    gru_vec.hip); h24: the MFMA whole-stack kernels specialised for H = 24 (gru_h24.hip), which the
    library itself picks above 2048 sequences; small: the general MFMA whole-stack kernels
    (gru_small.hip); per_layer: GEMM + recurrent kernels; per_layer_guarded: the same with the GEMM's
    guarded operand loads where the library would pick the unconditional ones."""
    for k in ("TGRU_FORCE_GENERIC", "TGRU_NO_H24", "TGRU_NO_VEC", "TGRU_GEMM_GUARDED"):
        monkeypatch.delenv(k, raising=False)
    if request.param.startswith("per_layer"):
        monkeypatch.setenv("TGRU_FORCE_GENERIC", "1")
        if request.param == "per_layer_guarded":
            monkeypatch.setenv("TGRU_GEMM_GUARDED", "1")
    elif request.param == "h24":
        monkeypatch.setenv("TGRU_NO_VEC", "1")
    elif request.param == "small":
        monkeypatch.setenv("TGRU_NO_VEC", "1")
        monkeypatch.setenv("TGRU_NO_H24", "1")
    import timegan_amd as tg
    tg.reload_switches()
    return request.param


@pytest.mark.parametrize("layout", ["tbf", "btf", "packed"])
@pytest.mark.parametrize("name", gru_case_names())
def test_golden_fixture(name, layout, kernel_path):
    case = load_gru_case(name)
    res = _run(case, layout)
    _compare(res, expected(case, "f32"), f"{name}[{layout}] vs torch-cpu fp32")
    _compare(res, expected(case, "f64"), f"{name}[{layout}] vs torch-cpu fp64")
