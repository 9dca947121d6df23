"""Benchmark of the GRU forward + BPTT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4] [--only]
                    [--no-cpu] [--no-training]

A step is one forward + full backward (dx and all weight gradients) of one 3-layer GRU stack over one
batch of synthetic input already resident in HBM, issued through the C ABI (libtgru.so).  With N > 1
every rank owns a batch shard of the same size (weak scaling) and the step ends with one RCCL
all-reduce of the flat weight-gradient bucket, as data-parallel training needs.

Rank 0 prints ONE JSON line.  The headline (`value`, `ms_per_step`, `config`, `roofline`, `cpu_baseline`,
`launches_per_step`) is the workload named by --workload (default c4: the per-GPU shard of the stress
configuration, the largest single-GPU configuration), timed for EXACTLY --steps steps after --warmup
warm-up steps between barrier + synchronize brackets.  Unless --only is given, `workloads` carries the
same record for each of c2 / c3 / c4 (the other two with step counts of their own), and
`timegan_training` the iterations per second of the three training phases on the HIP path at the c2 and c3
dimensions, next to the same trainer on torch's CPU nn.GRU at the c1 dimensions.

`value` is sequences/s over all ranks.  `roofline` prices the whole step (all kernels of forward +
backward, timed with HIP events on the launch stream) against the bound of the workload: HBM bytes for
H = 24, fp32 MFMA FLOPs for H >= 96.  `cpu_baseline` times the operator the reference itself calls
(torch.nn.GRU on CPU behind pack_padded_sequence) on the host cores of the same box, on a bounded sample of
the same workload, median of five steps per thread count; CPU model and core counts are stated.
If anything after the headline measurement stalls past TGRU_BENCH_DEADLINE seconds (default 600; 240 with several
ranks), rank 0
prints the line with what has been measured and `"stalled": true`, and every rank exits with status 3.
"""
import os

try:                                    # CPUs this process may use, read before OpenMP binds the main thread
    USABLE_CPUS = len(os.sched_getaffinity(0))
except AttributeError:
    USABLE_CPUS = os.cpu_count() or 1
os.environ.setdefault("OMP_PROC_BIND", "close")     # CPU legs: keep OpenMP threads on neighbouring cores

import argparse
import json
import statistics
import subprocess
import sys
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
# (steps, warmup) of a workload when it is not the headline
SUB_STEPS = {"c2": (200, 20), "c3": (100, 10), "c4": (10, 2)}
# features of the training probes at the workloads' dimensions (stock: 6, energy: 28)
TRAIN_FEATURES = {"c1": 5, "c2": 6, "c3": 28}
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


def host_cpu():
    """CPU model, physical / logical core counts of the machine and the cores this process may use."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    usable = USABLE_CPUS
    return {"model": model, "physical_cores": len(phys) or None, "logical_cpus": logical or (os.cpu_count() or 1),
            "usable_cpus": usable}


def cpu_thread_counts():
    """1 thread and the CPU share of a one-GPU box (16), bounded by what this process may use."""
    many = max(1, min(16, USABLE_CPUS))
    return [1] if many == 1 else [1, many]


def cpu_baseline(B, T, In, H, L, repeats=5):
    """torch.nn.GRU CPU fwd+bwd through pack/pad, as the reference calls it: median of `repeats` steps per
    thread count on a bounded sample of the workload (the stress shape: 256 sequences x the first 16 of the
    256 steps, the result scaled to full-length sequences -- the work is linear in T; a batch of 256 keeps the
    CPU's GEMMs fed, the figure is still a lower bound for the full batch of 2048)."""
    import torch.nn as nn
    results = []
    heavy = B * T * H > 1024 * 24 * 96
    Bc, Tc = (min(B, 256), min(T, 16)) if heavy else (B, T)
    if heavy:
        repeats = 3
    for threads in (cpu_thread_counts()[-1:] if heavy else cpu_thread_counts()):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        gru = nn.GRU(In, H, L, batch_first=True)
        x = torch.rand(Bc, Tc, In, requires_grad=True)
        lens = [Tc] * Bc

        def step():
            gru.zero_grad()
            p = nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=True)
            o, _ = gru(p)
            y, _ = nn.utils.rnn.pad_packed_sequence(o, batch_first=True, padding_value=0, total_length=Tc)
            y.sum().backward()
        for _ in range(2 if not heavy else 1):
            step()
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        results.append({"threads": threads, "seqs_per_s": Bc * (Tc / T) / med, "median_s": med})
    best = max(results, key=lambda r: r["seqs_per_s"])
    cpu = host_cpu()
    return {"value": round(best["seqs_per_s"], 2), "unit": "seqs/s", "cores": best["threads"],
            "kind": "reference", "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"],
            "logical_cpus": cpu["logical_cpus"], "usable_cpus": cpu["usable_cpus"],
            "sample": (f"torch {torch.__version__} CPU nn.GRU (the operator the reference calls) behind "
                       f"pack_padded_sequence/pad_packed_sequence, fwd+bwd, {Bc} sequences x {Tc} of {T} steps, "
                       f"same H/L, median of {repeats} steps after warm-up, OMP_PROC_BIND=close"
                       + (", scaled to full-length sequences (throughput per sequence depends on the batch: a lower "
                          "bound for the full batch)" if Tc != T else "") + "; thread counts tried: "
                       + ", ".join(f"{r['threads']}t={r['seqs_per_s']:.1f}/s" for r in results))}


def training_probe(tg, dev, features, H, L, T, B, group, graphs, out, iters=3, rnn_cls=None, count=False):
    """Secondary figure: 3-phase TimeGAN steps (TimeGANv1.py:573-812 loop bodies).
    One joint iteration = 10 x (generator step + embedder step) + 1 discriminator step
    = 116 GRU-stack forwards + 103 backwards plus the heads / losses / Adam work around them.
    Measured eagerly and (`graphs`) with each phase's step replayed as a HIP graph; with a
    process group the graphs contain the RCCL all-reduces.  Fills `out` key by key, so what has been
    measured survives if a later part does not finish.  `rnn_cls` (CPU leg): the recurrent module class,
    torch.nn.GRU for the reference's own operator."""
    cfg = tg.make_cfg(features=features, hidden=H, layers=L, seq_len=T, device=str(dev))
    x = torch.rand(B, T, features, device=dev)
    prefix = "hipgraph_" if graphs else ""
    torch.manual_seed(42)
    kw = {} if rnn_cls is None else {"rnn_cls": rnn_cls}
    tr = tg.TimeGANTrainer(cfg, device=dev, group=group, use_graphs=graphs, **kw)
    if graphs and not tr.use_graphs:
        out["hipgraph_note"] = "graph mode not available with this backend"
        return
    on_gpu = dev.type == "cuda"
    short = (group is not None and not graphs) or not on_gpu     # launch-bound or CPU-bound: keep it short
    for name, fn, n in (("embedding", tr.embedding_step, 10 if short else 20),
                        ("supervisor", tr.supervisor_step, 10 if short else 20),
                        ("joint", tr.joint_step, 2 if short else iters * (4 if graphs else 1))):
        for _ in range(3 if on_gpu else 1):
            fn(x)
        if on_gpu:
            torch.cuda.synchronize()
        n0 = tg.launch_count() if on_gpu else 0
        t0 = time.perf_counter()
        for _ in range(n):
            fn(x)
        if on_gpu:
            torch.cuda.synchronize()
        out[f"{prefix}{name}_iters_per_s"] = round(n / (time.perf_counter() - t0), 3)
        if on_gpu and not graphs:
            # the library's own launches (GRU, heads, losses, reductions, Adam); all GPU kernels of an iteration,
            # torch's elementwise / cat / fill kernels included, are counted by the profiler below
            out[f"{name}_tgru_launches_per_iter"] = round((tg.launch_count() - n0) / n, 1)
    if count:
        key = "joint_gpu_kernels_per_iter" if on_gpu else "joint_aten_ops_per_iter"
        out[("hipgraph_" if graphs else "") + key] = count_ops(lambda: tr.joint_step(x), on_gpu)


def count_ops(fn, device_kernels):
    """Runs fn() once under torch's profiler.  device_kernels: the number of GPU kernels it launched (torch's own
    elementwise / copy / fill kernels included, not only the library's); otherwise the number of ATen operator
    invocations on the CPU -- the CPU path's analogue of a launch count.  None if the profiler is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if device_kernels else [])
        with profile(activities=acts) as prof:
            fn()
            if device_kernels:
                torch.cuda.synchronize()
        n = 0
        for e in prof.events():
            is_dev = str(getattr(e, "device_type", "")).endswith("CUDA")
            if device_kernels and is_dev:
                n += 1
            elif not device_kernels and not is_dev and str(e.name).startswith("aten::"):
                n += 1
        return n
    except Exception as exc:       # noqa: BLE001 -- a diagnostic must not end the benchmark
        return f"unavailable: {type(exc).__name__}: {exc}"[:160]


def eager_module_call(tg, dev):
    """Host-side cost of the drop-in module in eager mode at the stock dimensions (B 128, T 24, H 24, L 3): one
    tg.GRU forward + backward per call pair, through the C++ autograd binding (the module's default) and through
    the ctypes binding; the kernels of a pair take about 41 us."""
    from timegan_amd import functional
    B, T, _, H, L, _ = WORKLOADS["c2"]
    gru = tg.GRU(H, H, L, batch_first=True).to(dev)
    x = torch.rand(B, T, H, device=dev, requires_grad=True)
    gy = torch.randn(B, T, H, device=dev)
    rec = {"config": f"tg.GRU({H}, {H}, {L}, batch_first=True) on ({B}, {T}, {H}): forward + backward, gradients reset each pair"}
    for name, cpp in (("cpp_binding_us", True), ("ctypes_binding_us", False)):
        if cpp and functional.cpp_op() is None:
            rec[name] = None
            continue
        gru.cpp_op = cpp

        def pair():
            gru.zero_grad(set_to_none=True)
            y, _ = gru(x)
            y.backward(gy)
        for _ in range(20):
            pair()
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            pair()
        torch.cuda.synchronize()
        rec[name] = round((time.perf_counter() - t0) / n * 1e6, 1)
    return rec


def generation_and_scoring(tg, dev, out, cpu_leg):
    """The water-quality configuration's path (6 features, T 24, H 24, 3 layers): generation = the inference path
    g(z) -> sup -> rec (TimeGANv1.py:556-570), three stack FORWARDS per sequence, and the post-hoc scorer
    (metrics/GeneralRNN.py:27-55 with the GRU cell, data flow of metrics/discriminative.py:57-78): a one-layer GRU
    classifier trained real-vs-synthetic, at the TimeGAN-paper width (In 6 -> H 24) and at the reference's own
    scorer config (In 28 -> H 256, discriminative.py:8-20)."""
    import numpy as np
    import torch.nn as nn
    F, H, L, T, N = 6, 24, 3, 24, 8000
    cfg = tg.make_cfg(features=F, hidden=H, layers=L, seq_len=T, device=str(dev))
    torch.manual_seed(42)
    tr = tg.TimeGANTrainer(cfg, device=dev)
    z = torch.rand(N, T, F, device=dev)
    for _ in range(3):
        tr.generate(z)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n0 = tg.launch_count()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        tr.generate(z)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    dev_ms = e0.elapsed_time(e1) / reps
    gru_bytes = 3 * N * 8.0 * L * T * H + 3 * 4.0 * L * (6 * H * H + 6 * H)      # forward only: read x, write y
    rec = {"config": f"generation: {N} sequences through g -> sup -> rec (3 stack forwards), F {F} H {H} L {L} T {T}",
           "value": round(N / wall, 1), "unit": "seqs/s", "ms_per_pass": round(1e3 * wall, 4),
           "device_ms_per_pass": round(dev_ms, 4), "tgru_launches_per_pass": (tg.launch_count() - n0) // reps,
           "roofline": {"bound": "hbm", "achieved": round(gru_bytes / (dev_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(gru_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes_per_pass": gru_bytes,
                        "note": "forward-only GRU bytes (8 L T H per sequence and stack) over the whole pass, heads included in the time"}}
    out["generation"] = rec
    rng = np.random.default_rng(0)
    scorers = {}
    out["scorer_training"] = scorers
    for name, (n, f, hidden) in {"in6_h24": (8000, 6, 24), "in28_h256": (1000, 28, 256)}.items():
        real = rng.uniform(0, 1, (n, T, f)).astype(np.float32)
        synth = rng.uniform(0, 1, (n, T, f)).astype(np.float32)
        tg.scorer.discriminative_score(real, synth, dev, hidden=hidden, steps=5, batch=128)      # warm-up
        torch.cuda.synchronize()
        steps = 100
        t0 = time.perf_counter()
        tg.scorer.discriminative_score(real, synth, dev, hidden=hidden, steps=steps, batch=128)
        torch.cuda.synchronize()
        r = {"config": f"GRU classifier In {f} -> H {hidden}, 1 layer, T {T}, minibatch 128, Adam; {n} + {n} windows",
             "value": round(steps / (time.perf_counter() - t0), 1), "unit": "training steps/s (evaluation of the test split included)"}
        if cpu_leg:
            torch.set_num_threads(cpu_thread_counts()[-1])
            t0 = time.perf_counter()
            tg.scorer.discriminative_score(real, synth, "cpu", hidden=hidden, steps=20, batch=128, rnn_cls=nn.GRU)
            r["cpu_baseline"] = {"value": round(20 / (time.perf_counter() - t0), 1), "unit": "training steps/s",
                                 "cores": cpu_thread_counts()[-1], "kind": "reference",
                                 "sample": "the same scorer on torch CPU nn.GRU, 20 steps"}
        scorers[name] = r
    if cpu_leg:
        torch.set_num_threads(cpu_thread_counts()[-1])
        cfgc = tg.make_cfg(features=F, hidden=H, layers=L, seq_len=T, device="cpu")
        trc = tg.TimeGANTrainer(cfgc, device="cpu", rnn_cls=nn.GRU)
        zc = torch.rand(2000, T, F)
        trc.generate(zc)
        t0 = time.perf_counter()
        trc.generate(zc)
        rec["cpu_baseline"] = {"value": round(2000 / (time.perf_counter() - t0), 1), "unit": "seqs/s",
                               "cores": cpu_thread_counts()[-1], "kind": "reference",
                               "sample": "the same inference path on torch CPU nn.GRU, 2000 sequences"}


def stress_training(tg, dev, out):
    """The three training phases at the stress configuration's dimensions (64 features, H 512, L 3, T 256) on one
    GPU: the per-GPU batch is the largest of 2048 / 1024 / 512 / 256 that the memory holds, judged from the peak of a
    joint iteration at batch 256 (everything a joint iteration keeps alive grows linearly with the batch)."""
    F, H, L, T = 64, 512, 3, 256
    cfg = tg.make_cfg(features=F, hidden=H, layers=L, seq_len=T, device=str(dev))
    torch.manual_seed(42)
    tr = tg.TimeGANTrainer(cfg, device=dev)
    total = torch.cuda.get_device_properties(dev).total_memory
    rec = {"config": f"{F} features, H {H}, L {L}, T {T}"}
    out["c4"] = rec

    def one(B, timed_joint):
        x = torch.rand(B, T, F, device=dev)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        res = {}
        for name, fn, n in (("embedding", tr.embedding_step, 2), ("supervisor", tr.supervisor_step, 2),
                            ("joint", tr.joint_step, timed_joint)):
            fn(x)                                   # warm-up (allocator growth, first launches)
            torch.cuda.synchronize()
            n0 = tg.launch_count()
            t0 = time.perf_counter()
            for _ in range(n):
                fn(x)
            torch.cuda.synchronize()
            res[f"{name}_iters_per_s"] = round(n / (time.perf_counter() - t0), 4)
            res[f"{name}_tgru_launches_per_iter"] = round((tg.launch_count() - n0) / n, 1)
        res["peak_memory_GiB"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)
        del x
        torch.cuda.empty_cache()
        return res

    small = one(256, 1)
    rec["batch_256"] = small
    per_seq = small["peak_memory_GiB"] * 2 ** 30 / 256
    B = next((b for b in (2048, 1024, 512) if per_seq * b < 0.85 * total), None)
    if B is not None:
        rec[f"batch_{B}"] = one(B, 1)
    rec["per_gpu_batch"] = B or 256
    rec["persist_timeouts"] = tg.persist_timeouts()
    rec["note"] = ("eager mode; one joint iteration = 84 stack forwards + 81 backwards at these dimensions, heads on the "
                   "library's GEMMs and wave-per-row LayerNorm kernels (no vendor GEMM, no torch LayerNorm)")



def cpu_training_baseline(tg, out):
    """The same trainer (modules, losses, optimiser stepping) on the CPU with torch.nn.GRU as its recurrent
    module -- the reference's own path (TimeGANv1.py:573-812) -- at the sine config (5 features, H 24, 3 layers,
    T 24, batch 128)."""
    import torch.nn as nn
    threads = cpu_thread_counts()[-1]
    torch.set_num_threads(threads)
    rec = {"note": f"build's trainer on torch CPU nn.GRU, sine config F5 H24 L3 T24 B128, {threads} threads "
                   "(like the GPU path it does not form the gradients a sub-step discards; the reference's loop does)"}
    out["cpu_c1"] = rec
    training_probe(tg, torch.device("cpu"), 5, 24, 3, 24, 128, None, False, rec, rnn_cls=nn.GRU, count=True)


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
    training_probe(tg, dev, TRAIN_FEATURES["c2"], H, L, T, B, dist.group.WORLD, True, out)
    device_barrier(dist, dev)
    print("DPGRAPH " + json.dumps(out), flush=True)
    dist.destroy_process_group()


def child_graph_probe(backend, rank, world, joint):
    """Graph replay with the RCCL all-reduces captured inside has not been exercised with more than one
    rank where this was written, so every rank runs it in a CHILD process (the children form their own
    process group on the next port): whatever happens there, the parent still reports what it has
    measured.  Two GPU processes per rank while it runs, so wide jobs skip it unless asked."""
    if os.environ.get("TGRU_BENCH_DP_GRAPH") != "1":
        joint["hipgraph_note"] = ("graph-mode data-parallel probe not run at %d ranks (a second process per GPU with a "
                                  "communicator of its own; set TGRU_BENCH_DP_GRAPH=1 to run it)" % world)
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


def measure_workload(tg, name, steps, warmup, dev, dist, rank, world, do_ar, use_graph=False):
    """Times `steps` steps of workload `name` after `warmup` warm-up steps.  Returns the record of the
    workload on rank 0 (None elsewhere): value, ms_per_step, roofline, launches_per_step."""
    B, T, In, H, L, desc = WORKLOADS[name]
    torch.manual_seed(42 + rank)
    weights = make_weights(In, H, L, dev, seed=42)           # replicated parameters
    x = torch.rand(T, B, In, device=dev)                      # U[0,1) like MinMax-scaled data
    dy = torch.randn(T, B, H, device=dev)
    runner = tg.GRURunner(B, T, In, H, L, tg.LAYOUT_TBF, dev)
    runner.set_weights(weights)

    for _ in range(warmup):
        runner.forward(x)
        runner.backward(dy)
        if do_ar:
            bucket_all_reduce(dist, runner.flat_grads)
    torch.cuda.synchronize()
    n0 = tg.launch_count()
    runner.forward(x)
    n1 = tg.launch_count()
    runner.backward(dy)
    n2 = tg.launch_count()
    launches = {"forward": n1 - n0, "backward": n2 - n1, "total": n2 - n0}
    torch.cuda.synchronize()
    # Optionally the step's launches are captured once into a HIP graph and the timed loop replays it; the
    # all-reduce of the data-parallel case stays outside the graph.
    graph = None
    if use_graph:
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
    grp = 10 if steps % 10 == 0 and H < 512 else 1
    ngrp = steps // grp
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(ngrp)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(ngrp)]
    t0 = time.perf_counter()
    for i in range(steps):
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
    nh = min(steps, 50)
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
    # the gradient all-reduce on its own (a zero bucket of the same size, so that the sums stay finite)
    ar_ms = None
    if do_ar:
        bucket = torch.zeros_like(runner.flat_grads)
        for _ in range(2):
            bucket_all_reduce(dist, bucket)
        device_barrier(dist, dev)
        n_ar = 20 if bucket.numel() < (1 << 22) else 5
        t1 = time.perf_counter()
        for _ in range(n_ar):
            bucket_all_reduce(dist, bucket)
        torch.cuda.synchronize()
        ar_ms = 1e3 * (time.perf_counter() - t1) / n_ar
    if rank != 0:
        return None

    flops, bytes_ = algorithmic_work(B, T, In, H, L)
    value = world * B * steps / elapsed
    if H < 64:
        achieved = bytes_ / (dev_ms_avg * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None}
    else:
        achieved = flops / (dev_ms_avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 5), "traffic": None}
    for tr_file in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_traffic_{name}.json") for r in (3, 2, 1)):
        if os.path.exists(tr_file):     # HBM bytes per step from separate rocprofv3 --pmc passes of this command
            with open(tr_file) as f:
                roof["traffic"] = json.load(f)["hbm_bytes_per_step"]
            roof["traffic_source"] = os.path.relpath(tr_file, ROOT)
            break
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
    return {"value": round(value, 2), "unit": "seqs/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 4),
            "config": {"workload": f"{name}: {desc}", "global_batch": world * B, "seq_len": T,
                       "launch": "host-issued kernels" if graph is None else "one HIP graph replay per step",
                       "parallelism": f"dp{world}" + (" + RCCL all-reduce of weight grads" if do_ar else "")},
            "roofline": roof, "launches_per_step": launches,
            "grad_bucket_bytes": int(runner.flat_grads.numel()) * 4,
            "allreduce_ms_per_step": None if ar_ms is None else round(ar_ms, 4)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the launcher.  It starts N
    copies of itself, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
    torch.distributed.run would set them), BEFORE anything here has touched a GPU, and never touches one itself.
    Rank 0's standard output (the one result line) is passed through; the other ranks' goes to standard error.
    The exit status is non-zero if any rank's is; when one rank fails the others are ended (by process id)."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with status {rc}; ending the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        if live:
            time.sleep(0.2)
    return status


def dry_run(args, world, rank, result_fd):
    """The launcher's plumbing without a GPU: process group over host memory, one all-reduce, one line."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    backend = "gloo" if args.backend == "nccl" and not torch.cuda.is_available() else args.backend
    got = 1
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.ones(4)
        dist.all_reduce(t)
        got = int(t[0].item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps({"metric": "GRU stack fwd+bwd sequences/sec", "dry_run": True, "n_gpus": world,
                                         "process_group_ranks": got,
                                         "backend": backend if world > 1 else None}) + "\n").encode())
    return 0


def main():
    if os.environ.get("TGRU_BENCH_CHILD") == "1":
        return dp_graph_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS), help="the headline workload")
    ap.add_argument("--only", action="store_true", help="measure the headline workload only")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines")
    ap.add_argument("--no-training", "--no-joint", dest="no_training", action="store_true",
                    help="skip the secondary 3-phase training probes")
    ap.add_argument("--no-allreduce", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay one captured HIP graph per step instead of issuing its launches from the host")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank code path (process group, all-reduces, graph capture of "
                         "them) with a single rank; set TIMEGAN_FORCE_COLLECTIVES=1 as well")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the ranks form the process group, all-reduce one host bucket and rank 0 "
                         "prints a line with n_gpus and the group size (checks the launcher where there is no GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)
    # The result line is the ONLY thing on stdout: libraries that print there (RCCL writes a version banner when
    # its first communicator is built) are sent to stderr at the file-descriptor level.
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, world, rank, result_fd)
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

    do_ar = multi and not args.no_allreduce
    head = measure_workload(tg, args.workload, args.steps, args.warmup, dev, dist, rank, world, do_ar, args.graph)
    out = None
    if rank == 0:
        out = {"metric": "GRU stack fwd+bwd sequences/sec", "value": head["value"], "unit": "seqs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": head["config"], "roofline": head["roofline"],
               "launches_per_step": head["launches_per_step"],
               # data-parallel facts of this run: the backend, the ranks of the process group that carried the
               # gradient all-reduce (RCCL ranks when the backend is nccl), the bucket and its time on its own
               "backend": args.backend if multi else None,
               "rccl_ranks": dist.get_world_size() if (multi and args.backend == "nccl") else 0,
               "process_group_ranks": dist.get_world_size() if multi else 1,
               "grad_bucket_bytes": head["grad_bucket_bytes"],
               "allreduce_ms_per_step": head["allreduce_ms_per_step"],
               "workloads": {args.workload: head}}

    printed = []

    def emit():
        """Rank 0 prints the one result line, once."""
        if rank == 0 and not printed:
            printed.append(True)
            os.write(result_fd, (json.dumps(out) + "\n").encode())

    # If anything below stalls (a wedged collective, a hung probe), the headline result still reaches the
    # driver: after the deadline rank 0 prints the line with what has been measured so far, marked as
    # stalled, and every rank leaves with a non-zero status -- a stall is a failed run.
    import threading

    def give_up():
        if rank == 0 and out is not None:
            out["stalled"] = True
            out["stalled_note"] = ("cut off at the deadline; the headline and every figure present were measured "
                                   "before the stall, which is in a secondary probe")
        emit()
        sys.stdout.flush()
        os._exit(3)

    # (with several ranks the secondary probes get four minutes: the headline is what a scaling run is for)
    limit = float(os.environ.get("TGRU_BENCH_DEADLINE", "240" if world > 1 else "600"))
    watchdog = threading.Timer(limit + (0 if rank == 0 else 5), give_up)
    watchdog.daemon = True
    watchdog.start()

    def secondary(what, fn):
        """A secondary probe must not cost the headline: with several ranks an exception in one of them is recorded
        in the line instead of ending the run (a single process lets it propagate: there it is a defect to see)."""
        if world == 1:
            return fn()
        try:
            return fn()
        except Exception as e:      # noqa: BLE001 -- whatever a collective or a probe raises
            if rank == 0:
                out.setdefault("secondary_errors", {})[what] = repr(e)[:300]
            return None

    if not args.only:
        for name in sorted(WORKLOADS):
            if name == args.workload:
                continue
            st, wu = SUB_STEPS[name]
            rec = secondary(f"workloads.{name}",
                            lambda: measure_workload(tg, name, st, wu, dev, dist, rank, world, do_ar))
            if rank == 0 and rec is not None:
                out["workloads"][name] = rec

    if not args.no_training and not args.only:
        training = {"note": "PyTorch autograd around the HIP GRU / head / loss kernels; iterations per second of the "
                            "embedding, supervisor and joint phases (one joint iteration = 10 x (G + E) + D "
                            "sub-steps), eager and with each step replayed as one HIP graph"}
        if rank == 0:
            out["timegan_training"] = training
        group = dist.group.WORLD if multi else None
        for name in ("c1", "c2", "c3"):
            B, T, _, H, L, _ = WORKLOADS["c2" if name == "c1" else name]
            rec = {"config": f"{TRAIN_FEATURES[name]} features, H {H}, L {L}, T {T}, batch {B} per GPU"}
            training[name] = rec
            secondary(f"timegan_training.{name}",
                      lambda: training_probe(tg, dev, TRAIN_FEATURES[name], H, L, T, B, group, False, rec, count=not multi))
            if multi and args.backend != "nccl":
                rec["hipgraph_note"] = "graph mode with collectives inside needs the RCCL backend"
            elif multi:
                if name == "c2":
                    child_graph_probe(args.backend, rank, world, rec)
            else:
                training_probe(tg, dev, TRAIN_FEATURES[name], H, L, T, B, group, True, rec)
        if not multi:
            stress_training(tg, dev, training)
        if rank == 0 and world == 1 and not args.no_cpu:
            cpu_training_baseline(tg, training)
    if not multi and not args.only:
        c5 = {}
        out["workloads"]["c5"] = c5
        generation_and_scoring(tg, dev, c5, not args.no_cpu)
        out["workloads"]["c2"]["eager_module_call"] = eager_module_call(tg, dev)

    if rank == 0 and world == 1 and not args.no_cpu:
        for name, rec in out["workloads"].items():
            if name not in WORKLOADS:
                continue
            B, T, In, H, L, _ = WORKLOADS[name]
            rec["cpu_baseline"] = cpu_baseline(B, T, In, H, L)
        out["cpu_baseline"] = out["workloads"][args.workload]["cpu_baseline"]
    emit()
    if dist:
        device_barrier(dist, dev)
        dist.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    sys.exit(main())
