"""Two independent GRU calls through tgru_forward_pair / tgru_backward_pair (one launch when both fit the
one-sequence-per-workgroup kernels) against the same two calls made separately: outputs, input gradients
and every weight gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(pair, rnn_a, x_a, len_a, rnn_b, x_b, len_b):
    from timegan_amd.gru import gru_pair
    for m in (rnn_a, rnn_b):
        m.zero_grad()
    xa = x_a.clone().requires_grad_(True)
    xb = x_b.clone().requires_grad_(True)
    if pair:
        ya, yb = gru_pair(rnn_a, xa, len_a, rnn_b, xb, len_b)
    else:
        ya, yb = rnn_a(xa, lengths=len_a)[0], rnn_b(xb, lengths=len_b)[0]
    g = torch.Generator(device="cuda:0").manual_seed(9)
    wa = torch.randn(ya.shape, device="cuda:0", generator=g)
    wb = torch.randn(yb.shape, device="cuda:0", generator=g)
    ((ya * wa).sum() + (yb * wb).sum()).backward()
    grads = [p.grad.clone() for m in (rnn_a, rnn_b) for p in m.parameters()]
    return [ya.detach(), yb.detach(), xa.grad, xb.grad] + grads


# (B_a, In_a, H_a, B_b, In_b, H_b, L_a, L_b, T, ragged)
CASES = [
    (128, 24, 24, 128, 24, 24, 3, 3, 24, False),     # embedder + generator of the stock config
    (128, 24, 24, 256, 24, 24, 3, 3, 24, False),     # recovery + discriminator on a doubled batch
    (40, 6, 24, 17, 30, 20, 2, 2, 9, True),          # different widths of the same class, ragged lengths
    (33, 24, 24, 12, 32, 32, 2, 2, 7, False),        # different padded width classes: two launches
    (20, 24, 24, 20, 24, 24, 3, 2, 8, False),        # different depths: two separate calls
    (16, 40, 40, 16, 24, 24, 2, 2, 6, False),        # one stack beyond the kernels' width: two calls
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "a%dx%dx%d_b%dx%dx%d_L%d%d_T%d_r%d" % tuple(int(v) for v in c))
def test_pair_equals_separate_calls(case):
    import timegan_amd as tg
    Ba, Ia, Ha, Bb, Ib, Hb, La, Lb, T, ragged = case
    torch.manual_seed(sum(int(v) for v in case))
    rnn_a = tg.GRU(Ia, Ha, num_layers=La, batch_first=True).to("cuda:0")
    rnn_b = tg.GRU(Ib, Hb, num_layers=Lb, batch_first=True).to("cuda:0")
    x_a = torch.randn(Ba, T, Ia, device="cuda:0")
    x_b = torch.randn(Bb, T, Ib, device="cuda:0")
    len_a = len_b = None
    if ragged:
        len_a = sorted((int(v) for v in torch.randint(1, T + 1, (Ba,))), reverse=True)
        len_b = sorted((int(v) for v in torch.randint(1, T + 1, (Bb,))), reverse=True)
    ref = _run(False, rnn_a, x_a, len_a, rnn_b, x_b, len_b)
    got = _run(True, rnn_a, x_a, len_a, rnn_b, x_b, len_b)
    for i, (u, v) in enumerate(zip(got, ref)):
        assert torch.equal(u, v), f"tensor {i}: max diff {float((u - v).abs().max()):.3e}"


def test_pair_with_a_frozen_network_fed_by_data():
    """The embedder / generator pair of a generator sub-step: the embedder is not updated and its input is
    data, so it owes nothing in the backward pass; the generator's gradients are those of the full pair."""
    import timegan_amd as tg
    from timegan_amd.modules import forward_pair
    cfg = tg.make_cfg(features=5, hidden=12, layers=2, seq_len=8, device="cuda:0")
    torch.manual_seed(3)
    nets = {k: v.to("cuda:0") for k, v in tg.build_networks(cfg).items()}
    x = torch.rand(16, 8, 5, device="cuda:0")
    z = torch.rand(16, 8, 5, device="cuda:0")
    t = [8] * 16

    def run(frozen):
        for k in ("emb", "g"):
            for p in nets[k].parameters():
                p.grad = None
                p.requires_grad_(not (frozen and k == "emb"))
        h, e_hat = forward_pair(nets["emb"], x, t, nets["g"], z, t)
        ((h * h).mean() + (e_hat * e_hat).mean() * 2 + (h * e_hat).mean()).backward()
        return [p.grad.clone() for p in nets["g"].parameters()]

    try:
        full, part = run(False), run(True)
        assert all(p.grad is None for p in nets["emb"].parameters())
    finally:
        for p in nets["emb"].parameters():
            p.requires_grad_(True)
    for u, v in zip(part, full):
        assert torch.equal(u, v)
