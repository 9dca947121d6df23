"""Benchmark of the GRU forward + BPTT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4] [--no-cpu]

A step is one forward + full backward (dx and all weight gradients) of one 3-layer GRU stack over one
batch of synthetic input already resident in HBM, issued through the C ABI (libtgru.so).  With N > 1
every rank owns a batch shard of the same size (weak scaling) and the step ends with one RCCL
all-reduce of the flat weight-gradient bucket, as data-parallel training needs.

Rank 0 prints ONE JSON line.  `value` is sequences/s over all ranks.  `roofline` prices the whole
step (all kernels of forward + backward, timed with HIP events on the launch stream) against the
bound of the workload: HBM bytes for H=24, fp32 MFMA FLOPs for H>=96.  `cpu_baseline` times the
operator the reference itself calls (torch.nn.GRU on CPU behind pack_padded_sequence) on the host
cores of the same box, on a bounded number of steps of the same workload.  The secondary training
probe (`timegan_training`) runs after the headline measurement under a deadline (TGRU_BENCH_DEADLINE,
150 s): if it stalls, the line is printed with what has been measured.
"""
import argparse
import json
import os
import sys
import subprocess
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (per-GPU batch, T, In(=H on the TimeGAN path), H, L, description)
    "c2": (128, 24, 24, 24, 3, "stock config: GRU stack fwd+bwd, B128 T24 H24 L3 (6 features -> H by the module MLP)"),
    "c3": (1024, 24, 96, 96, 3, "energy config: GRU stack fwd+bwd, B1024 T24 H96 L3"),
    "c4": (2048, 256, 512, 512, 3, "stress config: GRU stack fwd+bwd, per-GPU shard B2048 T256 H512 L3"),
}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak


def algorithmic_work(B, T, In, H, L):
    """FLOPs and bytes of one stack forward+backward (elementwise work excluded)."""
    flops = 0.0
    bytes_ = 0.0
    for l in range(L):
        in_l = In if l == 0 else H
        flops += B * T * (18.0 * H * in_l + 18.0 * H * H)          # fwd 6+6, bwd 12+12 per (in,H) pair
        bytes_ += B * T * 4.0 * (6 * H + 4 * in_l + 4 * H)           # fwd: x, y, 4 saved; bwd: dy, x, h, 4 saved, dx
        bytes_ += 12.0 * (3 * H * in_l + 3 * H * H + 6 * H)          # weights: read fwd, read bwd, write grads
    return flops, bytes_


def make_weights(In, H, L, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    k = 1.0 / (H ** 0.5)
    ws = []
    for l in range(L):
        in_l = In if l == 0 else H
        for shape in ((3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)):
            ws.append(((torch.rand(shape, generator=g) * 2 - 1) * k).to(device))
    return ws


def cpu_baseline(B, T, In, H, L, budget_s=20.0):
    """torch.nn.GRU CPU fwd+bwd through pack/pad, as the reference calls it."""
    import torch.nn as nn
    results = []
    for threads in (1, min(16, os.cpu_count() or 1)):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        gru = nn.GRU(In, H, L, batch_first=True)
        Bc = B if B * T * H <= 1024 * 24 * 96 else 256     # bounded sample for the stress shape
        x = torch.rand(Bc, T, In, requires_grad=True)
        lens = [T] * Bc

        def step():
            gru.zero_grad()
            p = nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=True)
            o, _ = gru(p)
            y, _ = nn.utils.rnn.pad_packed_sequence(o, batch_first=True, padding_value=0, total_length=T)
            y.sum().backward()
        step()
        n, t0 = 0, time.perf_counter()
        while True:
            step()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 200:
                break
        results.append({"threads": threads, "seqs_per_s": Bc * n / el, "steps": n, "batch": Bc})
        if threads == (os.cpu_count() or 1):
            break
    best = max(results, key=lambda r: r["seqs_per_s"])
    return {"value": round(best["seqs_per_s"], 2), "unit": "seqs/s", "cores": best["threads"],
            "kind": "reference",
            "sample": (f"torch {torch.__version__} CPU nn.GRU (the operator the reference calls) behind "
                       f"pack_padded_sequence/pad_packed_sequence, fwd+bwd, batch {best['batch']} x {best['steps']} "
                       f"steps of the same T/H/L; all thread counts tried: "
                       + ", ".join(f"{r['threads']}t={r['seqs_per_s']:.1f}" for r in results))}


def joint_training_probe(tg, dev, features, H, L, T, B, group, graphs, out, iters=3):
    """Secondary figure: 3-phase TimeGAN steps on the HIP GRU (TimeGANv1.py:573-812 loop bodies).
    One joint iteration = 10 x (generator step + embedder step) + 1 discriminator step
    = 116 GRU-stack forwards + 103 backwards plus the heads / losses / Adam work around them.
    Measured eagerly and (`graphs`) with each optimiser sub-step replayed as a HIP graph; with a
    process group the graphs contain the RCCL all-reduces.  Fills `out` key by key, so what has been
    measured survives if a later part does not finish."""
    cfg = tg.make_cfg(features=features, hidden=H, layers=L, seq_len=T, device=str(dev))
    x = torch.rand(B, T, features, device=dev)
    prefix = "hipgraph_" if graphs else ""
    torch.manual_seed(42)
    tr = tg.TimeGANTrainer(cfg, device=dev, group=group, use_graphs=graphs)
    if graphs and not tr.use_graphs:
        out["hipgraph_note"] = "graph mode not available with this backend"
        return
    short = group is not None and not graphs      # eager data-parallel probe: launch-bound, keep it short
    for name, fn, n in (("embedding", tr.embedding_step, 10 if short else 20),
                        ("supervisor", tr.supervisor_step, 10 if short else 20),
                        ("joint", tr.joint_step, 2 if short else iters * (4 if graphs else 1))):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn(x)
        torch.cuda.synchronize()
        out[f"{prefix}{name}_iters_per_s"] = round(n / (time.perf_counter() - t0), 3)


def device_barrier(dist, dev):
    """Barrier as an all-reduce of a device tensor: ordered with every other collective of this bench
    on the same device queue (a host-side barrier is not, with backends that run device-tensor
    collectives asynchronously), then drained."""
    t = torch.zeros(1, device=dev)
    bucket_all_reduce(dist, t)
    torch.cuda.synchronize()


def bucket_all_reduce(dist, t):
    """SUM over ranks, in place.  RCCL takes the device tensor as it is.  With the gloo backend (CPU
    rehearsals of the multi-rank flow, several ranks sharing one GPU) the tensor is staged through the host:
    gloo's own handling of device tensors stalls when the ranks' kernels queue up on one GPU."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        torch.cuda.synchronize()
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    else:
        dist.all_reduce(t)


def dp_graph_child():
    """Child of a multi-rank bench process: the graph-mode training probe with its own process group.
    Every rank prints 'DPGRAPH {json}' (the same figures: each rank times the same collective steps)."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "nccl"
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    import timegan_amd as tg
    B, T, _, H, L, _ = WORKLOADS["c2"]
    out = {}
    joint_training_probe(tg, dev, 6, H, L, T, B, dist.group.WORLD, True, out)
    device_barrier(dist, dev)
    print("DPGRAPH " + json.dumps(out), flush=True)
    dist.destroy_process_group()


def child_graph_probe(backend, rank, world, joint):
    """Graph replay with the RCCL all-reduces captured inside has not been exercised with more than one
    rank where this was written, so every rank runs it in a CHILD process (the children form their own
    process group on the next port): whatever happens there, the parent still reports what it has
    measured.  Two GPU processes per rank while it runs, so wide jobs skip it unless asked."""
    if world > 4 and os.environ.get("TGRU_BENCH_DP_GRAPH") != "1":
        joint["hipgraph_note"] = ("graph-mode data-parallel probe skipped at %d ranks "
                                  "(set TGRU_BENCH_DP_GRAPH=1 to run it)" % world)
        return
    env = dict(os.environ, TGRU_BENCH_CHILD="1",
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29541")) + 17))
    cmd = [sys.executable, "-u", os.path.abspath(__file__), "--backend", backend]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=90)
    except subprocess.TimeoutExpired:
        joint["hipgraph_note"] = "graph-mode data-parallel probe did not finish in 90 s"
        return
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DPGRAPH ")]
    if r.returncode == 0 and lines:
        joint.update(json.loads(lines[-1][len("DPGRAPH "):]))
    else:
        joint["hipgraph_note"] = ("graph-mode data-parallel probe failed (exit %d): %s"
                                  % (r.returncode, r.stderr.strip()[-200:]))


def main():
    if os.environ.get("TGRU_BENCH_CHILD") == "1":
        return dp_graph_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-allreduce", action="store_true")
    ap.add_argument("--no-joint", action="store_true", help="skip the secondary 3-phase training probe")
    ap.add_argument("--graph", action="store_true",
                    help="replay one captured HIP graph per step instead of issuing its three launches from the host "
                         "(measured slower here: 44.7 vs 41.3 us per step, the graph's inter-kernel gaps are larger)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank code path (process group, all-reduces, graph capture of "
                         "them) with a single rank; set TIMEGAN_FORCE_COLLECTIVES=1 as well")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("TGRU_BENCH_WATCHDOG"):       # debugging aid: dump every thread's stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["TGRU_BENCH_WATCHDOG"]), exit=True,
                                          file=open(f"gpurun_out/bench_watchdog_rank{rank}.log", "w"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one GPU per rank; the modulo only matters when ranks are rehearsed on fewer GPUs (gloo backend)
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if multi:
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    import timegan_amd as tg

    B, T, In, H, L, desc = WORKLOADS[args.workload]
    torch.manual_seed(42 + rank)
    weights = make_weights(In, H, L, dev, seed=42)           # replicated parameters
    x = torch.rand(T, B, In, device=dev)                      # U[0,1) like MinMax-scaled data
    dy = torch.randn(T, B, H, device=dev)
    runner = tg.GRURunner(B, T, In, H, L, tg.LAYOUT_TBF, dev)
    runner.set_weights(weights)
    do_ar = multi and not args.no_allreduce

    def step():
        runner.forward(x)
        runner.backward(dy)
        if do_ar:
            bucket_all_reduce(dist, runner.flat_grads)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # Optionally the step's launches (forward, backward, gradient reduce) are captured once into a HIP graph
    # and the timed loop replays it; the all-reduce of the data-parallel case stays outside the graph.
    graph = None
    if args.graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            runner.forward(x)
            runner.backward(dy)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    if dist:
        device_barrier(dist, dev)
    torch.cuda.synchronize()
    # HIP events bracket GROUPS of steps (an event record costs about as much as a launch): device time per
    # step = group time / group size, gaps between the step's kernels and between steps included (and, with
    # several ranks, the gradient all-reduce that ends each step)
    grp = 10 if args.steps % 10 == 0 else 1
    ngrp = args.steps // grp
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(ngrp)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(ngrp)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % grp == 0:
            ev0[i // grp].record()
        if graph is not None:
            graph.replay()
        else:
            runner.forward(x)
            runner.backward(dy)
        if i % grp == grp - 1:
            ev1[i // grp].record()
        if do_ar:
            bucket_all_reduce(dist, runner.flat_grads)
    torch.cuda.synchronize()
    if dist:
        device_barrier(dist, dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    dev_ms = sorted(a.elapsed_time(b) / grp for a, b in zip(ev0, ev1))
    dev_ms_avg = sum(dev_ms) / len(dev_ms)
    # the two halves of a step, timed apart in a short untimed pass of host-issued launches
    nh = min(args.steps, 50)
    eh = [[torch.cuda.Event(enable_timing=True) for _ in range(nh)] for _ in range(3)]
    for i in range(nh):
        eh[0][i].record()
        runner.forward(x)
        eh[1][i].record()
        runner.backward(dy)
        eh[2][i].record()
    torch.cuda.synchronize()
    fwd_ms_avg = sum(a.elapsed_time(b) for a, b in zip(eh[0], eh[1])) / nh
    bwd_ms_avg = sum(a.elapsed_time(b) for a, b in zip(eh[1], eh[2])) / nh

    out = None
    if rank == 0:
        flops, bytes_ = algorithmic_work(B, T, In, H, L)
        value = world * B * args.steps / elapsed
        if H < 64:
            achieved = bytes_ / (dev_ms_avg * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None}
        else:
            achieved = flops / (dev_ms_avg * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 5), "traffic": None}
        tr_file = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{args.workload}.json")
        if os.path.exists(tr_file):     # HBM bytes per step from separate rocprofv3 --pmc passes of this command
            with open(tr_file) as f:
                roof["traffic"] = json.load(f)["hbm_bytes_per_step"]
            roof["traffic_source"] = os.path.relpath(tr_file, ROOT)
        roof["kernel"] = "whole step: every kernel of one stack forward+backward (HIP events around the step)"
        roof["device_ms_per_step"] = round(dev_ms_avg, 4)
        roof["device_ms_median"] = round(dev_ms[len(dev_ms) // 2], 4)
        # the two halves of the step priced separately (forward kernel(s); backward + gradient reduce)
        wbytes = sum(4.0 * (3 * H * (In if l == 0 else H) + 3 * H * H + 6 * H) for l in range(L))
        fb = B * T * 4.0 * sum(6 * H for _ in range(L)) + wbytes            # x/y + 4 saved gate values
        bb = bytes_ - fb
        ff, bf = flops / 3.0, flops * 2.0 / 3.0
        roof["parts"] = {
            "forward": {"device_ms": round(fwd_ms_avg, 4), "GB/s": round(fb / fwd_ms_avg / 1e6, 2),
                        "TFLOP/s": round(ff / fwd_ms_avg / 1e9, 3)},
            "backward": {"device_ms": round(bwd_ms_avg, 4), "GB/s": round(bb / bwd_ms_avg / 1e6, 2),
                         "TFLOP/s": round(bf / bwd_ms_avg / 1e9, 3)}}
        roof["algorithmic_bytes_per_step"] = bytes_
        roof["algorithmic_flops_per_step"] = flops
        out = {
            "metric": "GRU stack fwd+bwd sequences/sec",
            "value": round(value, 2), "unit": "seqs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "global_batch": world * B, "seq_len": T,
                       "launch": "host-issued kernels" if graph is None else "one HIP graph replay per step",
                       "parallelism": f"dp{world}" + (" + RCCL all-reduce of weight grads" if do_ar else "")},
            "roofline": roof,
        }

    printed = []

    def emit():
        """Rank 0 prints the one result line, once."""
        if rank == 0 and not printed:
            printed.append(True)
            print(json.dumps(out), flush=True)

    # The secondary probes below run collectives that have only been rehearsed with one rank where this was
    # written.  If one of them stalls, the headline result must still reach the driver: after the deadline
    # rank 0 prints the line with what has been measured so far and every rank leaves.
    import threading

    def give_up():
        if rank == 0 and out is not None:
            out.setdefault("timegan_training", {})["note_deadline"] = "secondary probes cut off at the deadline"
        emit()
        sys.stdout.flush()
        os._exit(0)

    watchdog = threading.Timer(float(os.environ.get("TGRU_BENCH_DEADLINE", "150")) + (0 if rank == 0 else 5), give_up)
    watchdog.daemon = True
    watchdog.start()

    if not args.no_joint and args.workload == "c2":
        joint = {"note": "PyTorch autograd around the HIP GRU / head / loss kernels, batch %d per GPU, 6 features; "
                         "the eager figures are host-launch bound at this size" % B}
        if rank == 0:
            out["timegan_training"] = joint
        group = dist.group.WORLD if multi else None
        joint_training_probe(tg, dev, 6, H, L, T, B, group, False, joint)
        if multi and args.backend != "nccl":
            joint["hipgraph_note"] = "graph mode with collectives inside needs the RCCL backend"
        elif multi:
            child_graph_probe(args.backend, rank, world, joint)
        else:
            joint_training_probe(tg, dev, 6, H, L, T, B, group, True, joint)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(B, T, In, H, L)
    emit()
    if dist:
        device_barrier(dist, dev)
        dist.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    main()
