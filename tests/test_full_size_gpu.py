"""Parity at the benchmark configurations' real shapes.

* The energy stack at its real batch (B 1024, T 24, H 96, L 3) and the stress stack at its real depth
  (T 256, H 512, L 3; a batch of 256, and a ragged batch of 40) are
  compared with torch's CPU nn.GRU run live (a dependency present on the GPU box, not a reference file) at
  the standard bar: forward rtol 1e-5, gradients rtol 1e-4.
* The stress stack at its FULL per-GPU size (B 2048, 524 288 rows per weight-gradient reduction), which the
  CPU cannot finish in minutes, is checked through size-independent properties: a sequence's outputs and
  input gradients do not depend on its batch-mates, and the weight gradients of a batch are the sum of the
  weight gradients of its parts (each part small enough to be covered by the CPU comparison above).
* One embedding / supervisor / joint step of the trainer at the energy dimensions (28 features, H 96) against
  the same trainer on torch's CPU nn.GRU from identical parameters and noise.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close
from test_module_gpu import _compare_run, _pair

pytestmark = pytest.mark.gpu


def test_energy_stack_at_its_real_batch():
    ref, ours = _pair(96, 96, 3)
    x = torch.rand(1024, 24, 96, generator=torch.Generator().manual_seed(2))
    _compare_run(ref, ours, x, None, False, "c3 B1024 T24 H96 L3")


@pytest.mark.parametrize("B,lengths", [
    (256, None),
    (40, [256] * 8 + [255] * 8 + [200] * 8 + [77] * 8 + [3] * 4 + [1] * 4),
])
def test_stress_stack_at_its_real_depth(B, lengths):
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref, ours = _pair(512, 512, 3)
    x = torch.rand(B, 256, 512, generator=torch.Generator().manual_seed(3))
    _compare_run(ref, ours, x, lengths, False, f"c4 B{B} T256 H512 L3" + (" ragged" if lengths else ""))


def test_stress_stack_full_size_is_consistent_with_its_parts():
    """B 2048 x T 256 x H 512 x L 3 in one call against the same sequences in 8 calls of 256."""
    import timegan_amd as tg
    from bench import make_weights
    dev = torch.device("cuda:0")
    B, T, H, L, parts = 2048, 256, 512, 3, 8
    weights = make_weights(H, H, L, dev, seed=5)
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.rand(T, B, H, generator=g).to(dev)
    dy = torch.randn(T, B, H, generator=g).to(dev)
    full = tg.gru_forward_backward_raw(x, weights, B, T, tg.LAYOUT_TBF, dy)
    torch.cuda.synchronize()
    sums = [torch.zeros_like(w, dtype=torch.float64) for w in weights]
    nb = B // parts
    for p in range(parts):
        sl = slice(p * nb, (p + 1) * nb)
        part = tg.gru_forward_backward_raw(x[:, sl].contiguous(), weights, nb, T, tg.LAYOUT_TBF,
                                           dy[:, sl].contiguous())
        tag = f"part {p}"
        # per-sequence results: independent of the batch a sequence sits in
        assert_close(full["y"][:, sl].cpu().numpy(), part["y"].cpu().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " y")
        assert_close(full["h_n"][:, sl].cpu().numpy(), part["h_n"].cpu().numpy(), FWD_RTOL, FWD_ATOL_SCALE,
                     tag + " h_n")
        assert_close(full["dx"][:, sl].cpu().numpy(), part["dx"].cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE,
                     tag + " dx")
        for s, gpart in zip(sums, part["grads"]):
            s += gpart.double()
        del part
    names = ("dweight_ih", "dweight_hh", "dbias_ih", "dbias_hh")
    for i, (gfull, s) in enumerate(zip(full["grads"], sums)):
        assert_close(gfull.cpu().numpy(), s.cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE,
                     f"{names[i % 4]}_l{i // 4}: 524288-row reduction vs sum of 8 parts")


def test_three_phase_steps_at_the_energy_dimensions():
    """28 features, H 96, L 3, batch 64: one embedding, one supervisor and one joint step (10 x (G + E) + D)
    on the HIP path against the same trainer on torch's CPU nn.GRU, same parameters, same noise."""
    _three_phase_steps(28, 96, 3, 24, 64, 24, 64)


@pytest.mark.timeout(900)
def test_three_phase_steps_at_the_stress_dimensions(monkeypatch):
    """64 features, H 512, L 3, T 256 (the stress configuration) at batch 8: one embedding, one supervisor and
    one joint step on the HIP path -- persistent H > 128 recurrences, heads on the library's GEMMs and
    wave-per-row LayerNorms -- against the trainer on torch's CPU nn.GRU (about a minute of CPU time)."""
    import timegan_amd.head as head
    monkeypatch.setattr(head, "WIDE_LINEAR_MIN_ROWS", 1)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    _three_phase_steps(64, 512, 3, 256, 8, 256, 8)


def _three_phase_steps(F, H, L, T, B, T_joint, B_joint):
    import timegan_amd as tg
    cfg = tg.make_cfg(features=F, hidden=H, layers=L, seq_len=T, device="cpu")
    torch.manual_seed(21)
    cpu = tg.TimeGANTrainer(cfg, device="cpu", rnn_cls=nn.GRU)
    gpu = tg.TimeGANTrainer(cfg, device="cuda:0")
    gpu.load_state_dicts({k: {n: t.clone() for n, t in sd.items()} for k, sd in cpu.state_dicts().items()})
    g = torch.Generator().manual_seed(22)
    x = torch.rand(B, T, F, generator=g)
    noise = [torch.rand(B, T, F, generator=g) for _ in range(11)]

    def check_params(tag, rtol, atol_abs, worst=2.2, rare=5000):
        # Adam moves a parameter by about lr = 1e-3 per step whatever the gradient's size: a rounding-level
        # difference in a near-zero gradient shows up as a fraction of lr, hence the absolute term.  A network's
        # FIRST Adam step is exactly +-lr per element, so an element whose gradient is rounding noise around
        # zero can land 2 lr from the reference: at most one element in 5000 may do that, none may exceed it
        # (`worst` lr: a joint iteration updates a network ten times, and among the 786 432 elements of a 512-wide
        # weight matrix a few drift further: Adam turns ANY non-zero gradient into a step of about lr, so a
        # weight fed by a ReLU unit that is dead on the CPU and rounding-level alive on the GPU moves up to lr per
        # update, ten updates; `rare`: one element in that many may be out of tolerance).
        lr = 1e-3
        for k in cpu.nets:
            for (name, pc), (_, pg) in zip(cpu.nets[k].state_dict().items(), gpu.nets[k].state_dict().items()):
                got, ref = pg.cpu().numpy().astype(np.float64), pc.numpy().astype(np.float64)
                err = np.abs(got - ref)
                bad = err > atol_abs + rtol * np.abs(ref)
                assert bad.sum() <= max(1, bad.size // rare) and float(err.max(initial=0.0)) <= worst * lr + atol_abs, \
                    (f"{tag} {k}.{name}: {int(bad.sum())}/{bad.size} elements out of tolerance, "
                     f"max abs err {err.max():.3e}, max |ref| {np.abs(ref).max():.3e}")

    le_c, le_g = cpu.embedding_step(x), gpu.embedding_step(x.cuda())
    assert_close([le_g], [le_c], 1e-4, 0, "embedding loss")
    check_params("after embedding step", 1e-3, 1e-4)
    ls_c, ls_g = cpu.supervisor_step(x), gpu.supervisor_step(x.cuda())
    assert_close([ls_g], [ls_c], 1e-4, 0, "supervisor loss")
    check_params("after supervisor step", 1e-3, 1e-4)
    if (T_joint, B_joint) != (T, B):        # the joint step on shorter sequences (seq_len is read per call)
        for tr in (cpu, gpu):
            tr.seq_len = T_joint
            for net in tr.nets.values():
                net.seq_len = T_joint
        x = x[:B_joint, :T_joint].contiguous()
        noise = [z[:B_joint, :T_joint].contiguous() for z in noise]
    oc = cpu.joint_step(x, noise=[z.clone() for z in noise])
    og = gpu.joint_step(x.cuda(), noise=[z.cuda() for z in noise])
    assert_close(og["g_loss"], oc["g_loss"], 1e-3, 0, "joint g losses")
    assert_close(og["e_loss"], oc["e_loss"], 1e-3, 0, "joint e losses")
    assert_close([og["d_loss"]], [oc["d_loss"]], 1e-3, 0, "joint d loss")
    assert og["d_stepped"] == oc["d_stepped"]
    check_params("after joint step", 5e-3, 5e-4, worst=2.2 if H <= 128 else 10.5, rare=5000 if H <= 128 else 200)
    assert tg.persist_timeouts() == 0
