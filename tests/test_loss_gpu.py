"""Fused loss ops (libtgru.so tgru_loss_*) against the torch composition they replace
(the loss builders of outdated_files/TimeGANv1.py:410-553), values and gradients, on the GPU."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FWD_RTOL, GRAD_RTOL = 1e-5, 1e-4


def _close(a, b, rtol, what):
    scale = float(b.abs().max()) if b.numel() else 1.0
    assert torch.allclose(a, b, rtol=rtol, atol=rtol * max(scale, 1e-30)), \
        f"{what}: max err {float((a - b).abs().max()):.3e} (scale {scale:.3e})"


def _bce(y, one):
    return F.binary_cross_entropy_with_logits(y, torch.ones_like(y) if one else torch.zeros_like(y))


def _torch_loss(kind, ts, gamma):
    from timegan_amd import loss_ops as lo
    if kind == lo.EMBED_SIDE:
        x_tilde, x = ts
        return (10 * torch.sqrt(F.mse_loss(x_tilde, x)),)
    if kind == lo.SUPERVISOR:
        h_sup, h = ts
        return (F.mse_loss(h_sup[:, :-1, :], h[:, 1:, :]),)
    if kind == lo.EMBED_MAIN:
        x_tilde, x, h_sup, h = ts
        t0 = F.mse_loss(x_tilde, x)
        e0 = 10 * torch.sqrt(t0)
        return e0 + 0.1 * F.mse_loss(h_sup[:, :-1, :], h[:, 1:, :]), e0, t0
    if kind == lo.DISCRIMINATOR:
        y_real, y_fake, y_fake_e = ts
        return (_bce(y_fake, False) + _bce(y_real, True) + gamma * _bce(y_fake_e, False),)
    y_fake, y_fake_e, h_sup, h, x_hat, x = ts
    g_s = F.mse_loss(h_sup[:, :-1, :], h[:, 1:, :])
    v1 = torch.mean(torch.abs((x_hat.std(dim=0, unbiased=False) + 1e-6) - x.std(dim=0, unbiased=False) + 1e-6))
    v2 = torch.mean(torch.abs(x_hat.mean(dim=0) - x.mean(dim=0)))
    return (_bce(y_fake, True) + gamma * _bce(y_fake_e, True) + 100 * torch.sqrt(g_s) + 1000 * (v1 + v2),)


def _operands(kind, B, T, H, Fe, gen):
    from timegan_amd import loss_ops as lo
    dev = "cuda:0"
    r = lambda *s: torch.rand(*s, device=dev, generator=gen)
    n = lambda *s: torch.randn(*s, device=dev, generator=gen)
    if kind == lo.EMBED_SIDE:
        return [r(B, T, Fe), r(B, T, Fe)], [True, False]
    if kind == lo.SUPERVISOR:
        return [n(B, T, H), n(B, T, H)], [True, True]
    if kind == lo.EMBED_MAIN:
        return [r(B, T, Fe), r(B, T, Fe), n(B, T, H), n(B, T, H)], [True, False, True, True]
    if kind == lo.DISCRIMINATOR:
        return [3 * n(B, T, 1), 3 * n(B, T, 1), 3 * n(B, T, 1)], [True, True, True]
    return [3 * n(B, T, 1), 3 * n(B, T, 1), n(B, T, H), n(B, T, H), r(B, T, Fe), r(B, T, Fe)], \
        [True, True, True, True, True, False]


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("B,T,H,Fe", [(128, 24, 24, 6), (37, 5, 7, 3), (300, 12, 40, 28),
                                       (1024, 24, 96, 28),      # the energy config
                                       (72, 257, 512, 4)])      # index x width beyond 2^40: the 64-bit index path
def test_fused_loss_matches_torch(kind, B, T, H, Fe):
    from timegan_amd import loss_ops as lo
    gen = torch.Generator(device="cuda:0").manual_seed(100 * kind + B)
    ts, req = _operands(kind, B, T, H, Fe, gen)
    gamma = 0.7
    a = [t.clone().requires_grad_(q) for t, q in zip(ts, req)]
    b = [t.clone().requires_grad_(q) for t, q in zip(ts, req)]
    ref = _torch_loss(kind, a, gamma)
    got = lo.fused_loss(kind, b, B, T, H, Fe, gamma)
    for i, v in enumerate(ref):
        _close(got[i].detach(), v.detach(), FWD_RTOL, f"out[{i}]")
    # every output takes part in the backward, with different weights
    w = [1.0, 0.3, -2.0]
    sum(w[i] * v for i, v in enumerate(ref)).backward()
    sum(w[i] * got[i] for i in range(len(ref))).backward()
    for i, (u, v) in enumerate(zip(a, b)):
        if req[i]:
            _close(v.grad, u.grad, GRAD_RTOL, f"grad[{i}]")
        else:
            assert v.grad is None


def test_rank_shards_sum_to_whole_batch():
    """What two data-parallel ranks do -- partial sums per shard, added (the all-reduce), second pass
    about the global means, added again, final -- gives the whole batch's loss and gradients."""
    from timegan_amd import loss_ops as lo
    from timegan_amd._lib import check, lib
    B, T, H, Fe, gamma = 96, 10, 12, 5, 0.9
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    ts, req = _operands(lo.GENERATOR, B, T, H, Fe, gen)
    whole = [t.clone().requires_grad_(q) for t, q in zip(ts, req)]
    lo.fused_loss(lo.GENERATOR, whole, B, T, H, Fe, gamma)[0].backward()
    want = lo.fused_loss(lo.GENERATOR, ts, B, T, H, Fe, gamma)[0]

    lb = lib()
    lay = (ctypes.c_int64 * 4)()
    check(lb.tgru_loss_layout(lo.GENERATOR, T, Fe, lay), "layout")
    n_a, n_b, off_out, total = (int(v) for v in lay)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cut = 40   # unequal shards
    shards = [[t[:cut].contiguous() for t in ts], [t[cut:].contiguous() for t in ts]]
    bufs = [torch.zeros(total, device="cuda:0") for _ in shards]
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    for sh, buf in zip(shards, bufs):
        check(lb.tgru_loss_partial(lo.GENERATOR, 1, lo._ptrs(sh, 6), sh[0].shape[0], T, H, Fe, ptr(buf), st), "p1")
    tot = bufs[0][:n_a] + bufs[1][:n_a]
    for buf in bufs:
        buf[:n_a] = tot
    for sh, buf in zip(shards, bufs):
        check(lb.tgru_loss_partial(lo.GENERATOR, 2, lo._ptrs(sh, 6), sh[0].shape[0], T, H, Fe, ptr(buf), st), "p2")
    tot = bufs[0][n_a:n_a + n_b] + bufs[1][n_a:n_a + n_b]
    one = torch.ones(1, device="cuda:0")
    grads = []
    for sh, buf in zip(shards, bufs):
        buf[n_a:n_a + n_b] = tot
        check(lb.tgru_loss_final(lo.GENERATOR, sh[0].shape[0], T, H, Fe, gamma, ptr(buf), st), "final")
        _close(buf[off_out], want, FWD_RTOL, "sharded loss")
        g = [torch.empty_like(t) if q else None for t, q in zip(sh, req)]
        check(lb.tgru_loss_backward(lo.GENERATOR, lo._ptrs(sh, 6), lo._ptrs(g, 6), lo._ptrs([one, None, None], 3),
                                    sh[0].shape[0], T, H, Fe, gamma, ptr(buf), st), "bwd")
        grads.append(g)
    for i, q in enumerate(req):
        if q:
            _close(torch.cat([grads[0][i], grads[1][i]]), whole[i].grad, GRAD_RTOL, f"sharded grad[{i}]")


@pytest.mark.parametrize("n,first", [(1, 0), (7, 3), (4096, 0), (100003, 5), (24 * 6 * 300, 24 * 6 * 129)])
def test_counter_noise_is_bit_exact_and_shards_are_slices(n, first):
    """tgru_uniform against the CPU restatement (pinned by Philox's published known answers), bit for bit, at
    aligned and unaligned starts; a rank's rows are the slice of the tensor one process draws."""
    import oracle
    from timegan_amd.functional import uniform_
    seed, stream = 0x1234567890ABCDEF, 42
    out = torch.full((n + 8,), -1.0, device="cuda:0")
    uniform_(out[4:4 + n], seed, stream, first)
    got = out.cpu().numpy()
    assert (got[:4] == -1).all() and (got[4 + n:] == -1).all()          # nothing written outside the range
    assert np.array_equal(got[4:4 + n], oracle.uniform(n, seed, stream, first))
    odd = torch.empty(n + 1, device="cuda:0")[1:]                          # a start that is not 16-byte aligned
    uniform_(odd, seed, stream, first)
    assert np.array_equal(odd.cpu().numpy(), got[4:4 + n])


def test_trainer_noise_of_a_shard_is_the_slice_of_the_global_batch():
    """Two 'ranks' with shards of 5 and 3 rows of a batch of 8 draw, with the same seed, exactly the rows a single
    process draws for the whole batch (counter-based noise; here the ranks are played in one process)."""
    import timegan_amd as tg
    cfg = tg.make_cfg(features=5, hidden=12, layers=1, seq_len=6, device="cuda:0")
    x = torch.rand(8, 6, 5, device="cuda:0")
    torch.manual_seed(4)
    whole = tg.TimeGANTrainer(cfg, device="cuda:0", counter_noise=True)
    z = torch.stack(whole._noise_like(x, 3))
    assert float(z.min()) >= 0 and float(z.max()) < 1 and not torch.equal(z[0], z[1])
    parts = []
    for lo, hi in ((0, 5), (5, 8)):
        torch.manual_seed(4)
        tr = tg.TimeGANTrainer(cfg, device="cuda:0", counter_noise=True)
        tr._extent = (lo, hi, 8)
        parts.append(torch.stack(tr._noise_like(x[lo:hi], 3)))
    assert torch.equal(torch.cat(parts, dim=1), z)
