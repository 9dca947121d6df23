"""Host-side mirrors (modules, loss builders, three-phase steps) against fixtures captured from the
imported reference (tests/golden/make_golden.py).

On CPU the networks are built on torch.nn.GRU (`rnn_cls`), which checks the host logic alone; the
GPU variants of the same tests build them on the HIP GRU and check the whole product path.
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close

NETS = ("emb", "rec", "sup", "g", "d")


def _cfg(threshold=0.15, features=5):
    import timegan_amd as tg
    return tg.make_cfg(features=features, hidden=24, layers=3, seq_len=24, threshold=threshold, device="cpu")


def _load_sd(d, prefix, nets, device):
    for k in NETS:
        sd = {key[len(prefix) + len(k) + 1:]: torch.from_numpy(d[key]).to(device)
              for key in d.files if key.startswith(f"{prefix}{k}.")}
        nets[k].load_state_dict(sd)


def _build(device, on_gpu, features=5):
    import timegan_amd as tg
    kw = {} if on_gpu else {"rnn_cls": torch.nn.GRU}
    nets = tg.build_networks(_cfg(features=features), **kw)
    return {k: m.to(device) for k, m in nets.items()}


def _check_grads(d, tag, nets, rtol, atol_scale):
    for k in NETS:
        for name, p in nets[k].named_parameters():
            key = f"{tag}.{k}.{name}"
            if key not in d.files:
                continue
            got = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.cpu().numpy()
            assert_close(got, d[key], rtol, atol_scale, key)


# fixtures at the sine config's 5 features (c1) and the stock config's 6 features (c2)
FEATURES = {"c1": 5, "c2": 6}


def _modules(device, on_gpu, dims="c1"):
    d = np.load(os.path.join(GOLDEN, f"modules_{dims}.npz"))
    nets = _build(device, on_gpu, FEATURES[dims])
    _load_sd(d, "sd.", nets, device)
    for tag in ("full", "ragged"):
        t = [int(v) for v in d[f"{tag}.t"]]
        for k in NETS:
            nets[k].zero_grad()
            xin = torch.from_numpy(d[f"{tag}.{k}.in"]).to(device).requires_grad_(True)
            y = nets[k](xin, t)
            assert_close(y.detach().cpu().numpy(), d[f"{tag}.{k}.out"], 1e-5, 2e-6, f"{tag}.{k}.out")
            (y * torch.from_numpy(d[f"{tag}.{k}.gout"]).to(device)).sum().backward()
            assert_close(xin.grad.cpu().numpy(), d[f"{tag}.{k}.gin"], 1e-4, 1e-5, f"{tag}.{k}.gin")
            for name, p in nets[k].named_parameters():
                assert_close(p.grad.cpu().numpy(), d[f"{tag}.{k}.grad.{name}"], 1e-4, 1e-5,
                             f"{tag}.{k}.grad.{name}")


def _losses(device, on_gpu, dims="c1"):
    import timegan_amd as tg
    L = tg.losses
    d = np.load(os.path.join(GOLDEN, f"losses_{dims}.npz"))
    nets = _build(device, on_gpu, FEATURES[dims])
    _load_sd(d, "sd.", nets, device)
    x = torch.from_numpy(d["x"]).to(device)
    z = torch.from_numpy(d["z"]).to(device)
    t = [int(v) for v in d["t"]]
    n = nets

    def zero():
        for m in nets.values():
            m.zero_grad()

    zero()
    e0, xt = L.embedding_forward_side(n["emb"], n["rec"], x, t)
    e0.backward()
    assert_close(e0.item(), d["side.e_loss0"], 1e-5, 0, "side.e_loss0")
    assert_close(xt.detach().cpu().numpy(), d["side.x_tilde"], 1e-5, 2e-6, "side.x_tilde")
    _check_grads(d, "side.grad", nets, 1e-4, 2e-5)

    zero()
    e, e0, et0 = L.embedding_forward_main(n["emb"], n["sup"], n["rec"], x, t)
    e.backward()
    assert_close(e.item(), d["main.e_loss"], 1e-5, 0, "main.e_loss")
    assert_close(et0.item(), d["main.e_loss_t0"], 1e-5, 0, "main.e_loss_t0")
    _check_grads(d, "main.grad", nets, 1e-4, 2e-5)

    zero()
    s, h, hs = L.supervisor_forward(n["emb"], n["sup"], x, t)
    s.backward()
    assert_close(s.item(), d["sup.s_loss"], 1e-5, 0, "sup.s_loss")
    assert_close(hs.detach().cpu().numpy(), d["sup.h_sup"], 1e-5, 2e-6, "sup.h_sup")
    _check_grads(d, "sup.grad", nets, 1e-4, 2e-5)

    zero()
    dl = L.discriminator_forward(n["emb"], n["sup"], n["g"], n["d"], x, t, z)
    dl.backward()
    assert_close(dl.item(), d["disc.d_loss"], 1e-5, 0, "disc.d_loss")
    _check_grads(d, "disc.grad", nets, 1e-4, 2e-5)

    zero()
    gl = L.generator_forward(n["emb"], n["sup"], n["g"], n["d"], n["rec"], x, t, z)
    gl.backward()
    assert_close(gl.item(), d["gen.g_loss"], 1e-5, 0, "gen.g_loss")
    _check_grads(d, "gen.grad", nets, 1e-4, 2e-5)

    with torch.no_grad():
        xh = L.inference(n["sup"], n["g"], n["rec"], z, t)
    assert_close(xh.cpu().numpy(), d["inference.x_hat"], 1e-5, 2e-6, "inference.x_hat")


def _steps(device, on_gpu, use_graphs=False):
    """embedding x3, supervisor x3, joint x2 (D steps), joint x1 with a high threshold (D skipped)."""
    import timegan_amd as tg
    d = np.load(os.path.join(GOLDEN, "steps_c1.npz"))
    kw = {} if on_gpu else {"rnn_cls": torch.nn.GRU}
    tr = tg.TimeGANTrainer(_cfg(), device=device, use_graphs=use_graphs, **kw)
    assert tr.use_graphs == use_graphs
    _load_sd(d, "init.", tr.nets, device)

    def check_params(tag, rtol, atol_abs):
        # Adam moves every parameter by about lr = 1e-3 per step whatever the gradient's size, so a
        # rounding-level difference in a near-zero gradient shows up as a fraction of lr in the
        # parameter: the absolute tolerance is a fraction of lr, on top of a relative one.
        for k in NETS:
            for name, p in tr.nets[k].state_dict().items():
                assert_close(p.cpu().numpy(), d[f"{tag}.{k}.{name}"], rtol, 0.0, f"{tag}.{k}.{name}",
                             atol_abs=atol_abs)

    got = [tr.embedding_step(torch.from_numpy(xb).to(device)) for xb in d["embedding.x"]]
    assert_close(got, d["embedding.e_loss0"], 1e-4, 0, "embedding losses")
    check_params("after_embedding", 1e-3, 1e-4)

    got = [tr.supervisor_step(torch.from_numpy(xb).to(device)) for xb in d["supervisor.x"]]
    assert_close(got, d["supervisor.s_loss"], 1e-4, 0, "supervisor losses")
    check_params("after_supervisor", 1e-3, 1e-4)

    for tag in ("joint", "joint_skip_d"):
        tr.d_threshold = float(d[f"{tag}.threshold"])
        zs = [torch.from_numpy(z).to(device) for z in d[f"{tag}.z"]]
        g_all, e_all, d_all = [], [], []
        for i, xb in enumerate(d[f"{tag}.x"]):
            out = tr.joint_step(torch.from_numpy(xb).to(device), noise=zs[11 * i:11 * i + 11])
            g_all += out["g_loss"]; e_all += out["e_loss"]; d_all.append(out["d_loss"])
            assert out["d_stepped"] == (tag == "joint")
        # after tens of Adam steps losses are compared at 1e-3, parameters at half a learning rate
        assert_close(g_all, d[f"{tag}.g_loss"], 1e-3, 0, f"{tag} g losses")
        assert_close(e_all, d[f"{tag}.e_loss"], 1e-3, 0, f"{tag} e losses")
        assert_close(d_all, d[f"{tag}.d_loss"], 1e-3, 0, f"{tag} d losses")
        check_params(f"after_{tag}", 5e-3, 5e-4)


@pytest.mark.parametrize("dims", ["c1", "c2"])
def test_modules_match_reference_cpu(dims):
    _modules(torch.device("cpu"), False, dims)


@pytest.mark.parametrize("dims", ["c1", "c2"])
def test_losses_match_reference_cpu(dims):
    _losses(torch.device("cpu"), False, dims)


def test_training_steps_match_reference_cpu():
    _steps(torch.device("cpu"), False)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", ["c1", "c2"])
def test_modules_match_reference_gpu(dims):
    _modules(torch.device("cuda:0"), True, dims)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", ["c1", "c2"])
def test_losses_match_reference_gpu(dims):
    _losses(torch.device("cuda:0"), True, dims)


@pytest.mark.gpu
def test_training_steps_match_reference_gpu():
    _steps(torch.device("cuda:0"), True)


@pytest.mark.gpu
def test_training_steps_hip_graphs_match_reference_gpu():
    """Same fixture with every optimiser sub-step captured into a HIP graph after two eager calls."""
    _steps(torch.device("cuda:0"), True, use_graphs=True)


def _updates(device, all_gradients, hidden, rnn_cls=None, graphs=False):
    """Parameters after one supervisor step and one joint iteration from a fixed start and fixed noise."""
    import copy
    import timegan_amd as tg
    cfg = tg.make_cfg(features=5, hidden=hidden, layers=2, seq_len=8, batch_size=16, device=str(device))
    torch.manual_seed(5)
    kw = {} if rnn_cls is None else {"rnn_cls": rnn_cls}
    tr = tg.TimeGANTrainer(cfg, device=device, all_gradients=all_gradients, use_graphs=graphs, **kw)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(16, 8, 5, generator=g).to(device)
    noise = [torch.rand(16, 8, 5, generator=g).to(device) for _ in range(11)]
    out = [tr.supervisor_step(x), tr.embedding_step(x)]
    res = tr.joint_step(x, noise=noise)
    out += res["g_loss"] + res["e_loss"] + [res["d_loss"]]
    return out, {f"{k}.{n}": p.detach().clone() for k, m in tr.nets.items() for n, p in m.named_parameters()}


def test_discarded_gradients_do_not_change_the_updates_cpu():
    """The trainer leaves out the gradients of the networks a sub-step does not update; the losses and every
    parameter after the steps are those of a trainer that forms them all, as the reference does
    (TimeGANv1.py:765-790: zero_grad of all five networks, backward, step of two optimisers)."""
    dev = torch.device("cpu")
    la, pa = _updates(dev, True, 12, rnn_cls=torch.nn.GRU)
    lb, pb = _updates(dev, False, 12, rnn_cls=torch.nn.GRU)
    assert la == lb
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [12, 96])
def test_discarded_gradients_do_not_change_the_updates_gpu(hidden):
    """Same on the HIP path: dx-only backward calls (NULL gradient arrays / NULL dparams) and the halves of
    paired calls that owe nothing leave losses and parameters unchanged -- bit for bit at hidden size 12,
    where the same kernels run either way; to rounding at 96, where a bottom layer whose input wants no
    gradient runs the recurrent kernel without the input gradient inside (another summation order, and Adam
    turns a rounding difference in a tiny gradient into a difference of up to twice the learning rate)."""
    dev = torch.device("cuda:0")
    la, pa = _updates(dev, True, hidden)
    lb, pb = _updates(dev, False, hidden)
    if hidden == 12:
        assert la == lb
        for k in pa:
            assert torch.equal(pa[k], pb[k]), k
        return
    for u, v in zip(la, lb):
        assert abs(u - v) <= 1e-3 * abs(v), (u, v)
    for k in pa:
        d = (pa[k] - pb[k]).abs()
        assert float(d.mean()) < 2e-5 and float(d.max()) < 5e-3, (k, float(d.mean()), float(d.max()))


def test_same_weight_calls_stay_separate_from_two_gib_on():
    """Same-weight calls are evaluated as one concatenated batch only below 2 GiB of (rows, T, width) floats: the
    H > 128 kernels address a batch-major tensor through 32-bit offsets.  Meta tensors: no memory is touched."""
    from timegan_amd import losses
    T = 256
    lens = [T] * 2048
    h = torch.empty(2048, T, 512, device="meta")
    assert losses._batched([h, h], lens, 512) is None                     # 4096 x 256 x 512 floats = 2 GiB
    narrow = torch.empty(2048, T, 64, device="meta")
    assert losses._batched([narrow, narrow], lens, 512) is None           # the network's hidden size counts
    both = losses._batched([narrow, narrow], lens, 64)
    assert both is not None and both[0].shape[0] == 4096 and both[2] == [2048, 2048]
    small = torch.empty(1024, T, 512, device="meta")
    assert losses._batched([small, small, small], [T] * 1024, 512) is not None
    assert losses._batched([small, small], [T] * 1023 + [T - 1], 512) is None     # ragged lengths
