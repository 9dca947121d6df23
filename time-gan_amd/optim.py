"""Adam update of several small networks in one kernel launch (libtgru.so: tgru_adam_jobs).

`AdamStep` stands in for calling `.step()` on a few `torch.optim.Adam` instances one after the other:
it reads and updates THEIR state (`exp_avg`, `exp_avg_sq`, `step` of every parameter), so the
optimisers stay the owners of the moments, can still be stepped individually (another phase, the CPU
path) and their state_dicts stay what torch would have produced.  The arithmetic is torch's fused
Adam without weight decay / amsgrad / maximize, the only configuration the reference uses
(outdated_files/TimeGANv1.py:932-938: `Adam(params, lr=lr)`).
"""
import ctypes
from typing import Optional, Sequence

import torch
from torch.optim import Adam

from ._lib import MAX_ADAM_ELEMS, MAX_ADAM_JOBS, AdamJob, check, lib, on_device


def init_adam_state(opt: Adam) -> None:
    """Creates Adam's per-parameter state now instead of inside the first step(): the one-launch update
    reads it, and a step captured into a graph must not contain the zero-initialisation of its own
    moment estimates."""
    for group in opt.param_groups:
        saved = [p.grad for p in group["params"]]
        for p in group["params"]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt._init_group(group, [], [], [], [], [], [])
        for p, g in zip(group["params"], saved):
            p.grad = g


class AdamStep:
    def __init__(self, optimizers: Sequence[Adam], device):
        self.device = torch.device(device)
        self.entries = []            # (parameter, its state dict) of every optimiser, in order
        hyper = None
        for opt in optimizers:
            for group in opt.param_groups:
                h = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
                if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
                    raise ValueError("AdamStep covers plain Adam only (no weight decay, amsgrad or maximize)")
                if hyper is not None and h != hyper:
                    raise ValueError("AdamStep needs the same lr / betas / eps in every optimiser it steps")
                hyper = h
                for p in group["params"]:
                    st = opt.state[p]
                    if "exp_avg" not in st or not isinstance(st["step"], torch.Tensor) or not st["step"].is_cuda:
                        raise ValueError("AdamStep needs initialised state with step counters on the device "
                                         "(Adam(fused=True) + init_adam_state)")
                    self.entries.append((p, st))
        self.hyper = hyper
        self.optimizers = list(optimizers)
        # the one-launch kernel gives each tensor one workgroup: it is meant for many SMALL tensors
        self.fits = (len(self.entries) <= MAX_ADAM_JOBS
                     and all(p.numel() <= MAX_ADAM_ELEMS for p, _ in self.entries))
        self.jobs = (AdamJob * MAX_ADAM_JOBS)()

    def step(self, skip: Optional[torch.Tensor] = None) -> None:
        """One Adam update of every parameter that has a gradient; `skip`: device float tensor, a non-zero
        value leaves parameters, moments and step counters untouched (decided on the device)."""
        if not self.fits:           # large tensors: torch's own fused kernels, the skip flag as their found_inf
            for opt in self.optimizers:
                if skip is not None:
                    opt.found_inf = skip
                try:
                    opt.step()
                finally:
                    if skip is not None:
                        del opt.found_inf
            return
        n = 0
        for p, st in self.entries:
            g = p.grad
            if g is None:
                continue
            assert g.is_contiguous() and p.is_contiguous() and g.dtype == torch.float32
            j = self.jobs[n]
            j.param, j.grad = p.data_ptr(), g.data_ptr()
            j.exp_avg, j.exp_avg_sq, j.step = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
            j.n = p.numel()
            n += 1
        if n == 0:
            return
        lr, b1, b2, eps = self.hyper
        with on_device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            check(lib().tgru_adam_jobs(self.jobs, n, lr, b1, b2, eps,
                                       None if skip is None else ctypes.c_void_p(skip.data_ptr()), stream),
                  "tgru_adam_jobs")
