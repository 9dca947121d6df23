"""Seeded random shapes through the GRU module and the fused head against torch on the CPU: odd widths,
single steps, single sequences, ragged lengths, initial states -- whatever kernel family the shape selects."""
import os

import numpy as np
import pytest
import torch

from helpers import FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close

pytestmark = pytest.mark.gpu


def _case(rng):
    H = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 25, 31, 32, 33, 40, 48, 64, 96, 136, 160, 256]))
    In = int(rng.choice([1, 2, 5, 6, 7, 16, 24, 28, 31, 32, 33, 48]))
    L = int(rng.integers(1, 5))
    T = int(rng.choice([1, 2, 3, 7, 24, 33]))
    B = int(rng.choice([1, 2, 3, 17, 64, 130, 257]))
    return B, T, In, H, L, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TGRU_FUZZ_SEEDS", "24"))))   # more seeds on demand
def test_random_gru_shape_matches_torch(seed):
    import timegan_amd as tg
    rng = np.random.default_rng(1000 + seed)
    B, T, In, H, L, ragged, with_h0, batch_first = _case(rng)
    torch.manual_seed(seed)
    ref = torch.nn.GRU(In, H, num_layers=L, batch_first=batch_first)
    ours = tg.GRU(In, H, num_layers=L, batch_first=batch_first).to("cuda:0")
    ours.load_state_dict(ref.state_dict())
    x = torch.randn((B, T, In) if batch_first else (T, B, In))
    h0 = torch.randn(L, B, H) if with_h0 else None
    lens = sorted((int(v) for v in rng.integers(1, T + 1, B)), reverse=True) if ragged else [T] * B
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=batch_first, enforce_sorted=True)
    yp, hn_r = ref(packed, h0)
    y_r, _ = torch.nn.utils.rnn.pad_packed_sequence(yp, batch_first=batch_first, total_length=T)
    wy, wh = torch.randn(y_r.shape), torch.randn(hn_r.shape)
    ((y_r * wy).sum() + (hn_r * wh).sum()).backward()

    xo = x.clone().to("cuda:0").requires_grad_(True)
    y_o, hn_o = ours(xo, None if h0 is None else h0.to("cuda:0"), lengths=lens)
    ((y_o * wy.to("cuda:0")).sum() + (hn_o * wh.to("cuda:0")).sum()).backward()
    tag = f"B{B} T{T} In{In} H{H} L{L} ragged{ragged} h0{with_h0} bf{batch_first}"
    assert_close(y_o.detach().cpu().numpy(), y_r.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y " + tag)
    assert_close(hn_o.detach().cpu().numpy(), hn_r.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "h_n " + tag)
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx " + tag)
    for (name, p), q in zip(ref.named_parameters(), ours.parameters()):
        assert_close(q.grad.cpu().numpy(), p.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{name} {tag}")


def _large_case(rng):
    H = int(rng.choice([160, 192, 256, 384, 512]))       # 256 / 384 / 512: the backward kernel with resident W_hh
    In = int(rng.choice([20, 32, 40, 64, 96]))            # multiples of 32: the projection inside the forward kernel
    L = int(rng.integers(1, 3))
    T = int(rng.choice([1, 2, 3, 6]))
    B = int(rng.choice([1, 5, 31, 33, 63, 64, 65, 130, 300]))
    return B, T, In, H, L, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TGRU_FUZZ_SEEDS_LARGE", "16"))))
def test_random_large_hidden_shape_matches_torch(seed):
    """The same check on the persistent H > 128 kernels: batches that end inside a row tile or inside its first
    half, single steps, ragged lengths, initial states, both layouts."""
    import timegan_amd as tg
    rng = np.random.default_rng(9000 + seed)
    B, T, In, H, L, ragged, with_h0, batch_first = _large_case(rng)
    torch.manual_seed(seed)
    ref = torch.nn.GRU(In, H, num_layers=L, batch_first=batch_first)
    ours = tg.GRU(In, H, num_layers=L, batch_first=batch_first).to("cuda:0")
    ours.load_state_dict(ref.state_dict())
    x = torch.rand((B, T, In) if batch_first else (T, B, In))
    h0 = torch.randn(L, B, H) if with_h0 else None
    lens = sorted((int(v) for v in rng.integers(1, T + 1, B)), reverse=True) if ragged else [T] * B
    xr = x.clone().requires_grad_(True)
    hr = None if h0 is None else h0.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=batch_first, enforce_sorted=True)
    yp, hn_r = ref(packed, hr)
    y_r, _ = torch.nn.utils.rnn.pad_packed_sequence(yp, batch_first=batch_first, total_length=T)
    wy, wh = torch.randn(y_r.shape), torch.randn(hn_r.shape)
    ((y_r * wy).sum() + (hn_r * wh).sum()).backward()

    tg.persist_timeouts()
    xo = x.clone().to("cuda:0").requires_grad_(True)
    ho = None if h0 is None else h0.clone().to("cuda:0").requires_grad_(True)
    y_o, hn_o = ours(xo, ho, lengths=lens)
    ((y_o * wy.to("cuda:0")).sum() + (hn_o * wh.to("cuda:0")).sum()).backward()
    assert tg.persist_timeouts() == 0
    tag = f"B{B} T{T} In{In} H{H} L{L} ragged{ragged} h0{with_h0} bf{batch_first}"
    assert_close(y_o.detach().cpu().numpy(), y_r.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y " + tag)
    assert_close(hn_o.detach().cpu().numpy(), hn_r.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "h_n " + tag)
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx " + tag)
    if with_h0:
        assert_close(ho.grad.cpu().numpy(), hr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dh0 " + tag)
    for (name, p), q in zip(ref.named_parameters(), ours.parameters()):
        assert_close(q.grad.cpu().numpy(), p.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{name} {tag}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("TGRU_FUZZ_SEEDS", "24"))))
def test_random_head_shape_matches_torch(seed):
    import copy
    import torch.nn as nn
    from timegan_amd.head import head_apply, supported
    rng = np.random.default_rng(5000 + seed)
    deep = bool(rng.integers(0, 2))
    ln0 = bool(rng.integers(0, 2))
    # a LayerNorm over fewer than 8 features is ill-conditioned (rows of near-zero variance amplify rounding
    # by 1/sqrt(eps) in any implementation), so normalised widths start at 8; the others at 1
    din = int(rng.integers(8 if ln0 else 1, 33))
    dmid = int(rng.integers(8, 65)) if deep else int(rng.integers(1, 33))
    dout = int(rng.integers(1, 33)) if deep else dmid
    M = int(rng.choice([1, 2, 15, 16, 17, 100, 513, 3072]))
    want_dx = bool(rng.integers(0, 2))
    torch.manual_seed(seed)
    ref = {"ln0": nn.LayerNorm(din) if ln0 else None, "lin1": nn.Linear(din, dmid),
           "ln1": nn.LayerNorm(dmid) if deep else None, "lin2": nn.Linear(dmid, dout) if deep else None}
    for k in ("ln0", "ln1"):
        if ref[k] is not None:
            with torch.no_grad():
                ref[k].weight.uniform_(0.5, 1.5)
                ref[k].bias.uniform_(-0.5, 0.5)
    ours = {k: (None if v is None else copy.deepcopy(v).to("cuda:0")) for k, v in ref.items()}
    x, dy = torch.randn(M, din), torch.randn(M, dout)
    xr = x.clone().requires_grad_(want_dx)
    h = ref["lin1"](xr if ref["ln0"] is None else ref["ln0"](xr))
    y_r = h if not deep else ref["lin2"](torch.relu(ref["ln1"](h)))
    y_r.backward(dy)
    xo = x.clone().to("cuda:0").requires_grad_(want_dx)
    assert supported(xo, ours["lin1"], ours["lin2"])
    y_o = head_apply(xo, ours["ln0"], ours["lin1"], ours["ln1"], ours["lin2"])
    y_o.backward(dy.to("cuda:0"))
    tag = f"M{M} {din}->{dmid}->{dout} ln0{ln0} deep{deep}"
    assert_close(y_o.detach().cpu().numpy(), y_r.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y " + tag)
    if want_dx:
        assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx " + tag)
    for k in ref:
        if ref[k] is None:
            continue
        for (name, p), q in zip(ref[k].named_parameters(), ours[k].parameters()):
            assert_close(q.grad.cpu().numpy(), p.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{k}.{name} {tag}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("TGRU_FUZZ_SEEDS", "24"))))
def test_random_loss_shape_matches_torch(seed):
    """The fused loss ops against the torch composition on random shapes (test_loss_gpu.py holds the helpers)."""
    from test_loss_gpu import FWD_RTOL as LF, GRAD_RTOL as LG, _close, _operands, _torch_loss
    from timegan_amd import loss_ops as lo
    rng = np.random.default_rng(9000 + seed)
    kind = int(rng.integers(0, 5))
    B = int(rng.choice([1, 2, 3, 16, 17, 100, 128, 333]))
    T = int(rng.choice([2, 3, 7, 24, 40]))
    H = int(rng.integers(1, 49))
    Fe = int(rng.integers(1, 33))
    if kind == lo.GENERATOR and B < 3:
        B = 3            # the batch std of one or two rows is degenerate
    gamma = float(rng.uniform(0.1, 2.0))
    gen = torch.Generator(device="cuda:0").manual_seed(seed)
    ts, req = _operands(kind, B, T, H, Fe, gen)
    a = [t.clone().requires_grad_(q) for t, q in zip(ts, req)]
    b = [t.clone().requires_grad_(q) for t, q in zip(ts, req)]
    ref = _torch_loss(kind, a, gamma)
    got = lo.fused_loss(kind, b, B, T, H, Fe, gamma)
    tag = f"kind{kind} B{B} T{T} H{H} F{Fe}"
    for i, v in enumerate(ref):
        _close(got[i].detach(), v.detach(), LF, f"out[{i}] {tag}")
    sum(ref).backward()
    sum(got[i] for i in range(len(ref))).backward()
    for i, (u, v) in enumerate(zip(a, b)):
        if req[i]:
            _close(v.grad, u.grad, LG, f"grad[{i}] {tag}")
