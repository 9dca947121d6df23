"""Data-parallel path on CPU: 2 ranks over gloo must reproduce the single-process result on the
concatenated batch (losses, the D-step branch and the parameters after the step), including the
sqrt-of-global-mse and global-batch-moment terms.  The networks run on torch.nn.GRU here; the
collective logic under test is the same code that runs over RCCL on GPUs."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_trainer(group=None):
    import timegan_amd as tg
    torch.manual_seed(5)
    cfg = tg.make_cfg(features=5, hidden=12, layers=2, seq_len=8, device="cpu")
    return tg.TimeGANTrainer(cfg, device="cpu", rnn_cls=torch.nn.GRU, group=group)


def _data():
    g = torch.Generator().manual_seed(9)
    x = torch.rand(7, 8, 5, generator=g)           # global batch 7: shards 4 + 3 (world 2), 2 + 2 + 2 + 1 (world 4)
    noise = [torch.rand(7, 8, 5, generator=g) for _ in range(11)]
    return x, noise


def _generator_grads(tr, x, z):
    """Gradients of the generator loss (the one with sqrt-of-mse and batch-moment terms), summed
    over ranks, before any optimiser step."""
    from timegan_amd import losses, parallel
    n = tr.nets
    for m in n.values():
        m.zero_grad()
    loss = losses.generator_forward(n["emb"], n["sup"], n["g"], n["d"], n["rec"], x, tr.lengths_of(x), z,
                                    group=tr.group)
    loss.backward()
    parallel.all_reduce_grads(n.values(), group=tr.group)
    grads = {f"{k}.{name}": p.grad.clone() for k, m in n.items() for name, p in m.named_parameters()
             if p.grad is not None}
    return float(loss.detach()), grads


def _run_steps(tr, x, noise):
    out = {"emb": tr.embedding_step(x), "sup": tr.supervisor_step(x)}
    j = tr.joint_step(x, noise=noise)
    out.update(g=j["g_loss"], e=j["e_loss"], d=j["d_loss"], d_stepped=j["d_stepped"])
    return out


def _worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from timegan_amd import parallel
    tr = _make_trainer()
    x, noise = _data()
    xs = parallel.shard_batch(x)
    ns = [parallel.shard_batch(z) for z in noise]
    gl, gg = _generator_grads(tr, xs, ns[0])
    out = _run_steps(tr, xs, ns)
    if rank == 0:
        torch.save({"out": out, "sd": tr.state_dicts(), "g_loss0": gl, "g_grads0": gg}, path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_match_single_process(tmp_path, world):
    torch.set_num_threads(1)
    tr = _make_trainer()
    x, noise = _data()
    ref_gl, ref_gg = _generator_grads(tr, x, noise[0])
    ref = _run_steps(tr, x, noise)
    ref_sd = tr.state_dicts()

    path = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    got = torch.load(path, weights_only=False)

    # exactness of the sharded loss and of the rank-summed gradients, before Adam amplifies anything
    assert abs(got["g_loss0"] - ref_gl) <= 1e-5 * abs(ref_gl)
    assert set(got["g_grads0"]) == set(ref_gg)
    for name, g in ref_gg.items():
        scale = float(g.abs().max())
        np.testing.assert_allclose(got["g_grads0"][name].numpy(), g.numpy(), rtol=1e-4, atol=1e-5 * scale + 1e-9,
                                   err_msg=name)
    assert got["out"]["d_stepped"] == ref["d_stepped"]
    for k in ("emb", "sup"):      # first optimiser step of each phase: losses still agree tightly
        assert abs(got["out"][k] - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-6, k
    assert abs(got["out"]["d"] - ref["d"]) <= 2e-3 * abs(ref["d"])     # after 20 Adam steps
    np.testing.assert_allclose(got["out"]["g"], ref["g"], rtol=2e-3)
    np.testing.assert_allclose(got["out"]["e"], ref["e"], rtol=2e-3)
    for net, sd in ref_sd.items():
        for name, p in sd.items():
            np.testing.assert_allclose(got["sd"][net][name].numpy(), p.numpy(), rtol=5e-3, atol=5e-4,
                                       err_msg=f"{net}.{name}")


def test_shard_batch_covers_everything_once():
    from timegan_amd import parallel
    x = torch.arange(7)
    assert parallel.shard_batch(x).tolist() == list(range(7))      # world 1: identity


def _score_inputs():
    g = np.random.default_rng(3)
    t = np.linspace(0, 3, 9)[None, :, None]
    real = (np.sin(t * g.uniform(1, 2, (21, 1, 3)) + g.uniform(0, 3, (21, 1, 3))) + 1) / 2   # 21 windows: odd shards
    synth = g.uniform(0, 1, real.shape)
    return real.astype(np.float32), synth.astype(np.float32)


def _generate_and_score(group=None):
    """The generation + scoring flow of the water-quality config: n sequences from seeded noise, then the
    two post-hoc scores, all sharded over the ranks of the group."""
    import timegan_amd as tg
    tr = _make_trainer(group)
    x = tr.generate_all(7, seed=11, chunk=2)            # 7 sequences: shards of 4 and 3
    real, synth = _score_inputs()
    d = tg.scorer.discriminative_score(real, synth, "cpu", hidden=6, steps=12, batch=9, group=group,
                                       rnn_cls=torch.nn.GRU)
    p = tg.scorer.predictive_score(real, synth, "cpu", hidden=6, steps=12, batch=9, group=group,
                                   rnn_cls=torch.nn.GRU)
    return x, d, p


def _score_worker(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x, d, p = _generate_and_score(dist.group.WORLD)
    if rank == world - 1:          # the rank with the shorter shards reports: it must hold everything too
        torch.save({"x": x, "d": d, "p": p}, path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_generation_and_scores_match_single_process(tmp_path):
    torch.set_num_threads(1)
    x, d, p = _generate_and_score()
    assert x.shape == (7, 8, 5)
    path = str(tmp_path / "score.pt")
    mp.spawn(_score_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = torch.load(path, weights_only=False)
    np.testing.assert_allclose(got["x"].numpy(), x.numpy(), rtol=1e-5, atol=1e-6)
    assert abs(got["d"] - d) <= 1e-6            # accuracies are ratios of small integers
    assert abs(got["p"] - p) <= 1e-4 * abs(p) + 1e-6


class _DropShortBatches:
    """What a single process must see to mirror `world` ranks: a global batch with fewer rows than ranks is
    dropped by every rank (data.ResidentDataset.batches), so it is dropped here too."""

    def __init__(self, ds, world):
        self.ds, self.world = ds, world

    def batches(self, batch_size, shuffle=True, generator=None, rank=0, world=1):
        for b in self.ds.batches(batch_size, shuffle=shuffle, generator=generator):
            if b.shape[0] >= self.world:
                yield b


def _fit_small(group=None, phases=("embedding", "supervisor"), mirror_world=1):
    """fit() over a small resident dataset whose last global batch is ragged (11 windows, batches of 4:
    4 + 4 + 3)."""
    import timegan_amd as tg
    tr = _make_trainer(group)
    tr.cfg["system"]["batch_size"] = 4
    data = np.random.default_rng(4).uniform(0, 1, (11, 8, 5)).astype(np.float32)
    ds = tg.data.ResidentDataset(data, "cpu")
    if mirror_world > 1:
        ds = _DropShortBatches(ds, mirror_world)
    hist = tr.fit(ds, phases=phases, seed=6)
    return hist, tr.state_dicts()


def _fit_worker(rank, world, port, path, phases=("embedding", "supervisor")):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    hist, sd = _fit_small(dist.group.WORLD, phases)
    if rank == 0:
        torch.save({"hist": hist, "sd": sd}, path)
    dist.barrier()
    dist.destroy_process_group()


def test_fit_over_a_sharded_dataset_matches_single_process(tmp_path):
    """Every rank takes its shard of each global batch (ragged tail included) and runs the same number of
    steps; losses and parameters follow the single-process run on the whole batches."""
    torch.set_num_threads(1)
    ref_hist, ref_sd = _fit_small()
    path = str(tmp_path / "fit.pt")
    mp.spawn(_fit_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = torch.load(path, weights_only=False)
    for phase in ("embedding", "supervisor"):
        assert len(got["hist"][phase]) == len(ref_hist[phase]) == 3
        np.testing.assert_allclose(got["hist"][phase], ref_hist[phase], rtol=2e-3)
    for net in ("emb", "rec", "sup"):
        for name, p in ref_sd[net].items():
            np.testing.assert_allclose(got["sd"][net][name].numpy(), p.numpy(), rtol=5e-3, atol=5e-4, err_msg=f"{net}.{name}")


def test_fit_with_more_ranks_than_rows_in_the_tail_batch_and_the_joint_phase(tmp_path):
    """World size 4 over global batches of 4 + 4 + 3 windows: every rank holds ONE window of a full batch,
    the 3-window tail has fewer rows than ranks and is dropped by all of them alike, and all three phases
    (the joint phase with its rank-consistent discriminator branch included) follow the single-process run
    over the same batches."""
    torch.set_num_threads(1)
    phases = ("embedding", "supervisor", "joint")
    ref_hist, ref_sd = _fit_small(phases=phases, mirror_world=4)
    path = str(tmp_path / "fit4.pt")
    mp.spawn(_fit_worker, args=(4, _free_port(), path, phases), nprocs=4, join=True)
    got = torch.load(path, weights_only=False)
    for phase in ("embedding", "supervisor"):
        assert len(got["hist"][phase]) == len(ref_hist[phase]) == 2
        np.testing.assert_allclose(got["hist"][phase], ref_hist[phase], rtol=2e-3)
    assert len(got["hist"]["joint"]) == len(ref_hist["joint"]) == 2
    for a, b in zip(got["hist"]["joint"], ref_hist["joint"]):
        assert a["d_stepped"] == b["d_stepped"]
        np.testing.assert_allclose(a["g_loss"], b["g_loss"], rtol=5e-3)
        np.testing.assert_allclose(a["e_loss"], b["e_loss"], rtol=5e-3)
        np.testing.assert_allclose(a["d_loss"], b["d_loss"], rtol=5e-3)
    for net in ("emb", "rec", "sup", "g", "d"):
        for name, p in ref_sd[net].items():
            np.testing.assert_allclose(got["sd"][net][name].numpy(), p.numpy(), rtol=1e-2, atol=2e-3, err_msg=f"{net}.{name}")
