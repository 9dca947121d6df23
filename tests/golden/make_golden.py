"""Generates the golden fixtures under tests/golden/ (run once, in the build container).

    python tests/golden/make_golden.py [--ref /root/reference]

Only arrays travel: inputs, parameters and expected outputs.  Nothing from the reference tree is
copied.  Two sources are used:

  * GRU level (gru_*.npz): torch.nn.GRU on CPU (torch 2.10.0), called the way the reference
    calls it -- batch_first=True, pack_padded_sequence(enforce_sorted=True) in front,
    pad_packed_sequence(total_length=T) behind, hx never passed
    (outdated_files/TimeGANv1.py:35-38,56-65).  fp32 outputs are the parity target, fp64 outputs
    the tie-breaker.
  * module / loss / step level (modules_c1.npz, losses_c1.npz, steps_c1.npz; modules_c2.npz,
    losses_c2.npz at the stock config's 6 features): the reference's own
    classes and functions, imported from --ref with `wandb` (a logging service that is not
    installed) and the plotting call stubbed out.
  * data level (scorer_split.npz): the reference's SineWave and Energy dataset classes on a seeded
    draw / a small synthetic table, and the scorer's 75/25 split of that data.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------
# GRU level
# ----------------------------------------------------------------------------------------------
def run_torch_gru(x, weights, lengths, dy, dhn, dtype):
    B, T, In = x.shape
    L = len(weights)
    H = weights[0][1].shape[1]
    gru = nn.GRU(input_size=In, hidden_size=H, num_layers=L, batch_first=True).to(dtype)
    with torch.no_grad():
        for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(weights):
            getattr(gru, f"weight_ih_l{l}").copy_(torch.from_numpy(w_ih).to(dtype))
            getattr(gru, f"weight_hh_l{l}").copy_(torch.from_numpy(w_hh).to(dtype))
            getattr(gru, f"bias_ih_l{l}").copy_(torch.from_numpy(b_ih).to(dtype))
            getattr(gru, f"bias_hh_l{l}").copy_(torch.from_numpy(b_hh).to(dtype))
    xt = torch.from_numpy(x).to(dtype).requires_grad_(True)
    lens = [T] * B if lengths is None else [int(v) for v in lengths]
    packed = nn.utils.rnn.pack_padded_sequence(xt, lengths=lens, batch_first=True, enforce_sorted=True)
    out, h_n = gru(packed)
    y, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, padding_value=0, total_length=T)
    loss = (y * torch.from_numpy(dy).to(dtype)).sum()
    if dhn is not None:
        loss = loss + (h_n * torch.from_numpy(dhn).to(dtype)).sum()
    loss.backward()
    res = {"y": y.detach().numpy(), "h_n": h_n.detach().numpy(), "dx": xt.grad.numpy()}
    for l in range(L):
        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            res[f"d{nm}_l{l}"] = getattr(gru, f"{nm}_l{l}").grad.numpy()
    return res


GRU_CASES = {
    # name: (B, T, In, H, L, lengths, with_dhn)
    "gru_b4_t5_h24_l1": (4, 5, 24, 24, 1, None, False),
    "gru_b8_t24_h24_l3": (8, 24, 24, 24, 3, None, False),
    "gru_b8_t24_in6_h24_l2": (8, 24, 6, 24, 2, None, True),
    "gru_b4_t6_h96_l1": (4, 6, 96, 96, 1, None, False),
    "gru_b3_t9_h24_l3_ragged": (3, 9, 24, 24, 3, [7, 5, 2], True),
    "gru_b5_t6_in7_h10_l2_ragged": (5, 6, 7, 10, 2, [6, 6, 3, 2, 1], True),
    "gru_b20_t8_in28_h40_l2": (20, 8, 28, 40, 2, None, False),
}


def make_gru_fixtures():
    for idx, (name, (B, T, In, H, L, lengths, with_dhn)) in enumerate(GRU_CASES.items()):
        rng = np.random.default_rng(1000 + idx)
        k = 1.0 / np.sqrt(H)  # torch's default init range, rnn.py:296-299
        weights = []
        for l in range(L):
            in_l = In if l == 0 else H
            weights.append(tuple(rng.uniform(-k, k, size=s).astype(np.float32)
                                 for s in ((3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,))))
        x = rng.uniform(0, 1, size=(B, T, In)).astype(np.float32)
        dy = rng.standard_normal((B, T, H)).astype(np.float32)
        dhn = rng.standard_normal((L, B, H)).astype(np.float32) if with_dhn else None
        f32 = run_torch_gru(x, weights, lengths, dy, dhn, torch.float32)
        f64 = run_torch_gru(x, weights, lengths, dy, dhn, torch.float64)
        out = {"x": x, "dy": dy, "dims": np.array([B, T, In, H, L], dtype=np.int64)}
        if lengths is not None:
            out["lengths"] = np.array(lengths, dtype=np.int64)
        if dhn is not None:
            out["dhn"] = dhn
        for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(weights):
            out[f"weight_ih_l{l}"] = w_ih
            out[f"weight_hh_l{l}"] = w_hh
            out[f"bias_ih_l{l}"] = b_ih
            out[f"bias_hh_l{l}"] = b_hh
        for key, v in f32.items():
            out[f"f32_{key}"] = v.astype(np.float32)
        for key, v in f64.items():
            out[f"f64_{key}"] = v.astype(np.float64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ----------------------------------------------------------------------------------------------
# module / loss / step level, from the imported reference
# ----------------------------------------------------------------------------------------------
def import_reference(ref):
    sys.path.insert(0, ref)
    wandb = types.ModuleType("wandb")
    wandb.log = lambda *a, **k: None
    wandb.init = lambda *a, **k: None
    wandb.plot = types.SimpleNamespace(line_series=lambda *a, **k: None)
    sys.modules["wandb"] = wandb
    from outdated_files import TimeGANv1  # noqa
    TimeGANv1.visualisation.visualize = lambda *a, **k: None
    return TimeGANv1


def c1_cfg(F=5, H=24, L=3, T=24, threshold=0.15):
    return {
        "system": {"seq_len": T, "padding_value": 0.0, "lr": 1e-3, "batch_size": 1,
                   "device": "cpu", "jointly_num_epochs": 1, "dataset": "sine"},
        "emb": {"dim_features": F, "dim_hidden": H, "num_layers": L, "num_epochs": 1},
        "rec": {"dim_output": F, "dim_hidden": H, "num_layers": L},
        "sup": {"dim_hidden": H, "num_layers": L, "num_epochs": 1},
        "g": {"dim_latent": F, "dim_hidden": H, "num_layers": L},
        "d": {"dim_hidden": H, "num_layers": L, "threshold": threshold},
    }


def sd_arrays(prefix, module):
    return {f"{prefix}.{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grad_arrays(prefix, module):
    return {f"{prefix}.{k}": (p.grad.detach().numpy().copy() if p.grad is not None
                              else np.zeros(tuple(p.shape), np.float32))
            for k, p in module.named_parameters()}


def build_nets(R, cfg, seed):
    torch.manual_seed(seed)
    return {"emb": R.Embedding(cfg), "rec": R.Recovery(cfg), "sup": R.Supervisor(cfg),
            "g": R.Generator(cfg), "d": R.Discriminator(cfg)}


def make_module_fixtures(R, F=5, name="modules_c1", seed=7):
    """F2: every reference module, forward + backward, full and ragged lengths; C1 dims (5 features, the
    sine config) or C2 dims (6 features, the stock config)."""
    cfg = c1_cfg(F=F)
    B, T, H = 6, 24, 24
    nets = build_nets(R, cfg, seed)
    rng = np.random.default_rng(11 * seed)
    out = {}
    for k, m in nets.items():
        out.update(sd_arrays("sd." + k, m))
    inputs = {"emb": rng.uniform(0, 1, (B, T, F)), "g": rng.uniform(0, 1, (B, T, F)),
              "rec": rng.standard_normal((B, T, H)), "sup": rng.standard_normal((B, T, H)),
              "d": rng.standard_normal((B, T, H))}
    for tag, t in (("full", [T] * B), ("ragged", [24, 20, 20, 11, 3, 1])):
        out[f"{tag}.t"] = np.array(t, dtype=np.int64)
        for k, m in nets.items():
            m.zero_grad()
            xin = torch.tensor(inputs[k], dtype=torch.float32, requires_grad=True)
            y = m(xin, t)
            gy = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float32)
            (y * gy).sum().backward()
            out[f"{tag}.{k}.in"] = xin.detach().numpy().copy()
            out[f"{tag}.{k}.out"] = y.detach().numpy().copy()
            out[f"{tag}.{k}.gout"] = gy.numpy().copy()
            out[f"{tag}.{k}.gin"] = xin.grad.numpy().copy()
            out.update(grad_arrays(f"{tag}.{k}.grad", m))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def make_loss_fixtures(R, F=5, name="losses_c1", seed=11):
    """F3: the five loss builders (TimeGANv1.py:410-553) and the parameter grads they produce; C1 dims
    (5 features) or C2 dims (6 features)."""
    cfg = c1_cfg(F=F)
    B, T = 16, 24
    nets = build_nets(R, cfg, seed)
    rng = np.random.default_rng(111 if seed == 11 else 101 * seed)
    x = torch.tensor(rng.uniform(0, 1, (B, T, F)), dtype=torch.float32)
    z = torch.tensor(rng.uniform(0, 1, (B, T, F)), dtype=torch.float32)
    t = [T] * B
    out = {"x": x.numpy().copy(), "z": z.numpy().copy(), "t": np.array(t, dtype=np.int64)}
    for k, m in nets.items():
        out.update(sd_arrays("sd." + k, m))

    def zero():
        for m in nets.values():
            m.zero_grad()

    def grads(tag):
        for k, m in nets.items():
            out.update(grad_arrays(f"{tag}.grad.{k}", m))

    zero()
    e_loss0, _x = R._embedding_forward_side(emb=nets["emb"], rec=nets["rec"], x=x, t=t)
    e_loss0.backward()
    out["side.e_loss0"] = e_loss0.detach().numpy().copy()
    out["side.x_tilde"] = _x.detach().numpy().copy()
    grads("side")

    zero()
    e_loss, e_loss0, e_loss_t0 = R._embedding_forward_main(emb=nets["emb"], sup=nets["sup"],
                                                            rec=nets["rec"], x=x, t=t)
    e_loss.backward()
    out["main.e_loss"] = e_loss.detach().numpy().copy()
    out["main.e_loss0"] = e_loss0.detach().numpy().copy()
    out["main.e_loss_t0"] = e_loss_t0.detach().numpy().copy()
    grads("main")

    zero()
    s_loss, h, h_sup = R._supervisor_forward(emb=nets["emb"], sup=nets["sup"], x=x, t=t)
    s_loss.backward()
    out["sup.s_loss"] = s_loss.detach().numpy().copy()
    out["sup.h"] = h.detach().numpy().copy()
    out["sup.h_sup"] = h_sup.detach().numpy().copy()
    grads("sup")

    zero()
    d_loss = R._discriminator_forward(emb=nets["emb"], sup=nets["sup"], g=nets["g"], d=nets["d"],
                                      x=x, t=t, z=z)
    d_loss.backward()
    out["disc.d_loss"] = d_loss.detach().numpy().copy()
    grads("disc")

    zero()
    g_loss = R._generator_forward(emb=nets["emb"], sup=nets["sup"], g=nets["g"], d=nets["d"],
                                  rec=nets["rec"], x=x, t=t, z=z)
    g_loss.backward()
    out["gen.g_loss"] = g_loss.detach().numpy().copy()
    grads("gen")

    with torch.no_grad():
        out["inference.x_hat"] = R._inference(sup=nets["sup"], g=nets["g"], rec=nets["rec"],
                                              z=z, t=t).numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def make_step_fixtures(R):
    """F4: the reference's three trainers run on fixed batches and fixed noise.

    embedding_trainer (3 batches), supervisor_trainer (3 batches), joint_trainer (2 batches with
    threshold 0.15 -> D steps; 1 more batch with threshold 100 -> D step skipped), Adam lr 1e-3
    (TimeGANv1.py:573-812, optimisers as built at :933-938).
    """
    from torch.optim import Adam
    B, T, F = 16, 24, 5
    rng = np.random.default_rng(222)
    out = {}
    cfg = c1_cfg()
    nets = build_nets(R, cfg, 13)
    for k, m in nets.items():
        out.update(sd_arrays("init." + k, m))
    lr = cfg["system"]["lr"]
    opts = {"emb_side": Adam(nets["emb"].parameters(), lr=lr),
            "emb_main": Adam(nets["emb"].parameters(), lr=lr),
            "rec": Adam(nets["rec"].parameters(), lr=lr),
            "sup": Adam(nets["sup"].parameters(), lr=lr),
            "g": Adam(nets["g"].parameters(), lr=lr),
            "d": Adam(nets["d"].parameters(), lr=lr)}
    real_samples = [rng.uniform(0, 1, (T, F)) for _ in range(60)]

    # record losses returned by the reference's loss builders
    log = {"side": [], "sup": [], "gen": [], "main": [], "disc": []}

    def wrap(name, key, pick):
        orig = getattr(R, name)

        def f(*a, **k):
            r = orig(*a, **k)
            log[key].append(float(pick(r).detach()))
            return r
        setattr(R, name, f)
        return orig

    originals = {
        "_embedding_forward_side": wrap("_embedding_forward_side", "side", lambda r: r[0]),
        "_supervisor_forward": wrap("_supervisor_forward", "sup", lambda r: r[0]),
        "_generator_forward": wrap("_generator_forward", "gen", lambda r: r),
        "_embedding_forward_main": wrap("_embedding_forward_main", "main", lambda r: r[0]),
        "_discriminator_forward": wrap("_discriminator_forward", "disc", lambda r: r),
    }

    # fixed noise: every rand_like on a training-batch-shaped tensor pops the next z
    z_queue = []
    orig_rand_like = torch.rand_like

    def rand_like(inp, *a, **k):
        if tuple(inp.shape) == (B, T, F) and z_queue:
            return z_queue.pop(0).clone()
        return orig_rand_like(inp, *a, **k)
    torch.rand_like = rand_like

    def batches(n):
        return [torch.tensor(rng.uniform(0, 1, (B, T, F)), dtype=torch.float64) for _ in range(n)]

    def snapshot(tag):
        for k, m in nets.items():
            out.update(sd_arrays(f"{tag}.{k}", m))

    try:
        xb = batches(3)
        out["embedding.x"] = np.stack([b.numpy() for b in xb]).astype(np.float32)
        R.embedding_trainer(emb=nets["emb"], sup=nets["sup"], rec=nets["rec"],
                            emb_opt=opts["emb_side"], rec_opt=opts["rec"], dl=xb, cfg=cfg,
                            real_samples=real_samples)
        out["embedding.e_loss0"] = np.array(log["side"])
        snapshot("after_embedding")

        xb = batches(3)
        out["supervisor.x"] = np.stack([b.numpy() for b in xb]).astype(np.float32)
        R.supervisor_trainer(emb=nets["emb"], sup=nets["sup"], rec=nets["rec"],
                             sup_opt=opts["sup"], dl=xb, cfg=cfg, real_samples=real_samples)
        out["supervisor.s_loss"] = np.array(log["sup"])
        snapshot("after_supervisor")

        for tag, nb, thr in (("joint", 2, 0.15), ("joint_skip_d", 1, 100.0)):
            cfg_j = c1_cfg(threshold=thr)
            xb = batches(nb)
            zs = [torch.tensor(rng.uniform(0, 1, (B, T, F)), dtype=torch.float32)
                  for _ in range(11 * nb)]
            z_queue[:] = [z.clone() for z in zs]
            for k in ("gen", "main", "disc"):
                log[k].clear()
            out[f"{tag}.x"] = np.stack([b.numpy() for b in xb]).astype(np.float32)
            out[f"{tag}.z"] = np.stack([z.numpy() for z in zs])
            R.joint_trainer(emb=nets["emb"], sup=nets["sup"], g=nets["g"], d=nets["d"],
                            rec=nets["rec"], g_opt=opts["g"], d_opt=opts["d"],
                            sup_opt=opts["sup"], rec_opt=opts["rec"], emb_opt=opts["emb_main"],
                            dl=xb, real_samples=[s.copy() for s in real_samples], cfg=cfg_j)
            assert not z_queue, "noise queue not consumed as expected"
            out[f"{tag}.g_loss"] = np.array(log["gen"])
            out[f"{tag}.e_loss"] = np.array(log["main"])
            out[f"{tag}.d_loss"] = np.array(log["disc"])
            out[f"{tag}.threshold"] = np.array(thr)
            snapshot(f"after_{tag}")
    finally:
        torch.rand_like = orig_rand_like
        for name, fn in originals.items():
            setattr(R, name, fn)
    path = os.path.join(HERE, "steps_c1.npz")
    np.savez_compressed(path, **out)
    print(f"steps_c1: {os.path.getsize(path) / 1024:.0f} KiB")


def make_data_fixtures(ref):
    """F5: the reference's data classes (data/SineWave.py, data/Energy.py) on a seeded draw and a
    small synthetic table, and the arrays of the 75/25 split that metrics/discriminative.py:57-78
    builds from them (that script cannot run as a whole -- it needs trained checkpoints -- so the
    split is taken with the same index expressions on the reference-generated data)."""
    import tempfile
    sys.path.insert(0, ref)
    from data import SineWave, Energy
    out = {}
    np.random.seed(7)
    ds = SineWave.SineWave(samples_number=40, seq_len=24, features_dim=6)
    real = np.asarray(ds.get_distribution())
    out["sine.seed"] = np.array(7)
    out["sine.data"] = real
    rng = np.random.RandomState(11)
    table = rng.rand(60, 4) * np.array([1.0, 10.0, 100.0, 0.5]) + np.array([0.0, -3.0, 50.0, 2.0])
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
        f.write("a,b,c,d\n")
        for row in table:
            f.write(",".join(repr(float(v)) for v in row) + "\n")
        path = f.name
    en = Energy.Energy(seq_len=24, path=path)
    os.unlink(path)
    out["table"] = table
    out["energy.windows"] = np.asarray(en.get_distribution())
    t, msl = Energy.extract_time(np.asarray(en.get_distribution())[:5])
    out["energy.time"] = np.array(t)
    out["energy.max_seq_len"] = np.array(msl)
    # the split (discriminative.py:57-78) on real = sine data, synthetic = a seeded stand-in
    sd = np.random.RandomState(5).rand(*real.shape)
    n_events = len(real)
    idx = np.arange(n_events)
    n_train = int(.75 * n_events)
    train_idx, test_idx = idx[:n_train], idx[n_train:]
    out["split.synth"] = sd
    out["split.X_stock_train"] = real[train_idx, :24, :]
    out["split.X_synth_train"] = sd[train_idx, :24, :]
    out["split.X_stock_test"] = real[test_idx, :24, :]
    out["split.y_stock_test"] = real[test_idx, -1, :]
    out["split.y_stock_train"] = real[train_idx, -1, :]
    out["split.y_synth_train"] = sd[train_idx, -1, :]
    path = os.path.join(HERE, "scorer_split.npz")
    np.savez_compressed(path, **out)
    print(f"scorer_split: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="all", choices=["all", "gru", "ref", "ref_c2", "data"])
    args = ap.parse_args()
    torch.set_num_threads(1)
    if args.only in ("all", "gru"):
        make_gru_fixtures()
    if args.only in ("all", "ref"):
        R = import_reference(args.ref)
        make_module_fixtures(R)
        make_loss_fixtures(R)
        make_step_fixtures(R)
    if args.only in ("all", "ref_c2"):      # the stock config's width: 6 features
        R = import_reference(args.ref)
        make_module_fixtures(R, F=6, name="modules_c2", seed=13)
        make_loss_fixtures(R, F=6, name="losses_c2", seed=17)
    if args.only in ("all", "data"):
        make_data_fixtures(args.ref)
