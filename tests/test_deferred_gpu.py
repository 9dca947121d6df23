"""deferred_reductions(): the partial-gradient sums of a whole backward pass in one launch must leave
exactly the gradients of the ordinary pass -- also when a network is used twice (its second gradient is
ADDED to the first, so the first must be complete by then) and when some parameters want no gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(hidden=12):
    import timegan_amd as tg
    cfg = tg.make_cfg(features=5, hidden=hidden, layers=2, seq_len=8, device="cuda:0")
    torch.manual_seed(21)
    return {k: m.to("cuda:0") for k, m in tg.build_networks(cfg).items()}


def _grads(nets, deferred, twice, freeze, zero=True):
    from timegan_amd.deferred import deferred_reductions
    from timegan_amd.modules import forward_pair
    for m in nets.values():
        if zero:
            m.zero_grad()
        for p in m.parameters():
            p.requires_grad_(True)
    if freeze:
        for p in nets["rec"].parameters():
            p.requires_grad_(False)
    g = torch.Generator(device="cuda:0").manual_seed(4)
    x = torch.rand(16, 8, 5, device="cuda:0", generator=g)
    z = torch.rand(16, 8, 5, device="cuda:0", generator=g)
    t = [8] * 16
    h, e_hat = forward_pair(nets["emb"], x, t, nets["g"], z, t)
    if twice:      # the supervisor and the recovery network each see two inputs in SEPARATE calls
        out = [nets["sup"](h, t), nets["sup"](e_hat, t), nets["rec"](h, t), nets["rec"](e_hat, t)]
    else:
        out = list(forward_pair(nets["rec"], h, t, nets["sup"], h, t))
    loss = sum((o * o).mean() * (i + 1) for i, o in enumerate(out))
    if deferred:
        with deferred_reductions():
            loss.backward()
    else:
        loss.backward()
    return {f"{k}.{n}": (None if p.grad is None else p.grad.clone()) for k, m in nets.items() for n, p in m.named_parameters()}


@pytest.mark.parametrize("twice", [False, True])
@pytest.mark.parametrize("freeze", [False, True])
@pytest.mark.parametrize("hidden", [12, 96])      # 96: heads with the weights in LDS, four jobs per head
def test_deferred_equals_immediate(twice, freeze, hidden):
    nets = _nets(hidden)
    ref = _grads(nets, False, twice, freeze)
    got = _grads(nets, True, twice, freeze)
    assert ref.keys() == got.keys()
    for k in ref:
        if ref[k] is None:
            assert got[k] is None, k
        else:
            assert torch.equal(got[k], ref[k]), f"{k}: max diff {float((got[k] - ref[k]).abs().max()):.3e}"


@pytest.mark.parametrize("set_to_none", [True, False])
def test_deferred_passes_accumulate_into_existing_gradients(set_to_none):
    """Two backward passes without zeroing in between (gradient accumulation), and a pass over gradients
    zeroed in place (zero_grad(set_to_none=False)): autograd ADDS an arriving gradient to an existing one at
    once, so such a gradient must not be deferred."""
    nets = _nets()
    once = _grads(nets, False, False, False)
    if set_to_none:
        _grads(nets, True, False, False)                       # first pass fills .grad
        got = _grads(nets, True, False, False, zero=False)     # second pass adds to it
        want = {k: None if v is None else 2 * v for k, v in once.items()}
    else:
        for m in nets.values():
            m.zero_grad(set_to_none=False)                     # .grad stays a tensor of zeros
        got = _grads(nets, True, False, False, zero=False)
        want = once
    for k in want:
        if want[k] is None:
            assert got[k] is None, k
        else:
            assert torch.allclose(got[k], want[k], rtol=1e-6, atol=0), \
                f"{k}: max diff {float((got[k] - want[k]).abs().max()):.3e}"
