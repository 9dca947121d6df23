"""Worker of test_dp_graph_gpu.py (run as a script, one process, one GPU).

Trains a small TimeGAN three ways from the same seed and noise -- plain single-process eager; the
data-parallel code path (every collective forced to run on a one-rank RCCL group) eager; and that
path with each sub-step captured into a HIP graph, collectives included -- and checks that losses
and final parameters agree."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timegan_amd as tg                      # noqa: E402
from timegan_amd import parallel              # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1, device_id=dev)
    cfg = tg.make_cfg(features=5, hidden=12, layers=2, seq_len=8, device="cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.rand(16, 8, 5, device=dev, generator=gen)
    noise = [[torch.rand(x.shape, device=dev, generator=gen) for _ in range(11)] for _ in range(4)]

    def run(force, graphs):
        parallel.FORCE_COLLECTIVES = force
        torch.manual_seed(3)
        tr = tg.TimeGANTrainer(cfg, device=dev, group=dist.group.WORLD if force else None, use_graphs=graphs)
        assert tr.use_graphs == graphs
        log = [tr.embedding_step(x) for _ in range(4)] + [tr.supervisor_step(x) for _ in range(4)]
        for i in range(4):
            o = tr.joint_step(x, noise=noise[i])
            log += o["g_loss"] + o["e_loss"] + [o["d_loss"]]
        torch.cuda.synchronize()
        return torch.tensor(log), tr.state_dicts()

    ref_log, ref_sd = run(False, False)
    for name, (force, graphs) in {"dp_eager": (True, False), "dp_graph": (True, True)}.items():
        log, sd = run(force, graphs)
        assert torch.allclose(log, ref_log, rtol=2e-4, atol=1e-6), \
            f"{name}: losses differ, max {float((log - ref_log).abs().max()):.3e}"
        for net in ref_sd:
            for k, v in ref_sd[net].items():
                w = sd[net][k]
                assert torch.allclose(w, v, rtol=1e-3, atol=2e-5), \
                    f"{name}: {net}.{k} differs, max {float((w - v).abs().max()):.3e}"
        print(name, "ok", flush=True)

    # the data-parallel gradient exchange is in place: every gradient of a step is a view of its optimiser group's
    # arena, which is what gets all-reduced (nothing is concatenated or copied back)
    parallel.FORCE_COLLECTIVES = True
    tr = tg.TimeGANTrainer(cfg, device=dev, group=dist.group.WORLD)
    tr.joint_step(x, noise=noise[0])
    tr.embedding_step(x)
    arena = tr._arenas[("emb", "rec")]
    grads = [p.grad for k in ("emb", "rec") for p in tr.nets[k].parameters()]
    assert arena.n > 0 and all(g is not None and arena.owns(g) for g in grads)
    assert set(tr._arenas) >= {("g", "sup"), ("emb", "rec")}
    print("arena ok", flush=True)

    # generation + post-hoc scores: the sharded code path (all-gather, sharded minibatches, all-reduced
    # gradients and counts over RCCL) against the plain one
    import numpy as np
    rng = np.random.default_rng(3)
    real = rng.uniform(0, 1, (41, 8, 5)).astype(np.float32)
    synth = rng.uniform(0, 1, (41, 8, 5)).astype(np.float32)

    def score(force):
        parallel.FORCE_COLLECTIVES = force
        group = dist.group.WORLD if force else None
        torch.manual_seed(3)
        tr = tg.TimeGANTrainer(cfg, device=dev, group=group)
        xg = tr.generate_all(37, seed=5, chunk=16)
        d = tg.scorer.discriminative_score(real, synth, dev, hidden=8, steps=10, batch=16, group=group)
        p = tg.scorer.predictive_score(real, synth, dev, hidden=8, steps=10, batch=16, group=group)
        return xg, d, p

    x0, d0, p0 = score(False)
    x1, d1, p1 = score(True)
    assert x0.shape == (37, 8, 5) and torch.allclose(x0, x1, rtol=1e-5, atol=1e-6)
    assert abs(d0 - d1) <= 1e-6 and abs(p0 - p1) <= 1e-4 * abs(p0) + 1e-6, (d0, d1, p0, p1)
    print("generate/score ok", flush=True)
    dist.destroy_process_group()
    print("WORKER_OK", flush=True)


if __name__ == "__main__":
    main()
