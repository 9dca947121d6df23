"""GPU tests of the nn.GRU drop-in module: PackedSequence in/out, autograd, sizes beyond the golden
fixtures checked live against torch's CPU nn.GRU (a dependency present on the GPU box, not a
reference file), and the shapes the reference's own smoke functions use."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close

pytestmark = pytest.mark.gpu


def _pair(In, H, L, seed=0):
    import timegan_amd as tg
    torch.manual_seed(seed)
    ref = nn.GRU(In, H, L, batch_first=True)
    ours = tg.GRU(In, H, L, batch_first=True)
    ours.load_state_dict(ref.state_dict())
    return ref, ours.to("cuda:0")


def _compare_run(ref, ours, x, lengths, packed, tag):
    B, T, _ = x.shape
    g = torch.Generator().manual_seed(1)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().to("cuda:0").requires_grad_(True)
    lens = [T] * B if lengths is None else lengths
    pr = nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=True, enforce_sorted=True)
    yr, hr = ref(pr)
    yr, _ = nn.utils.rnn.pad_packed_sequence(yr, batch_first=True, padding_value=0, total_length=T)
    if packed:
        po = nn.utils.rnn.pack_padded_sequence(xo, lens, batch_first=True, enforce_sorted=True)
        yo, ho = ours(po)
        assert isinstance(yo, nn.utils.rnn.PackedSequence)
        yo, _ = nn.utils.rnn.pad_packed_sequence(yo, batch_first=True, padding_value=0, total_length=T)
    else:
        yo, ho = ours(xo, lengths=lengths)
    gy = torch.randn(yr.shape, generator=g)
    gh = torch.randn(hr.shape, generator=g)
    ((yr * gy).sum() + (hr * gh).sum()).backward()
    ((yo * gy.cuda()).sum() + (ho * gh.cuda()).sum()).backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, f"{tag} y")
    assert_close(ho.detach().cpu().numpy(), hr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, f"{tag} h_n")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} dx")
    for (name, pr_), (_, po_) in zip(ref.named_parameters(), ours.named_parameters()):
        assert_close(po_.grad.cpu().numpy(), pr_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} d{name}")


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("B,T,In,H,L,lengths", [
    (128, 24, 24, 24, 3, None),                       # C1/C2 stack
    (9, 7, 6, 24, 2, [7, 7, 6, 4, 4, 3, 2, 1, 1]),    # ragged, In != H, whole-stack kernels
    (64, 24, 96, 96, 3, None),                        # C3 stack (W_hh resident in LDS)
    (20, 6, 28, 256, 1, None),                        # scorer shape: In 28 -> H 256, one layer
    (17, 5, 100, 100, 2, [5, 5, 5, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1, 1, 1, 1]),   # H=100: odd tile count
    (16, 4, 64, 512, 2, None),                        # C4 sizes, small batch (W_hh streamed from L2)
    (300, 12, 24, 24, 3, None),                       # one-sequence-per-workgroup kernels, backward workgroups loop over 2 sequences
    (600, 5, 6, 24, 2, [5] * 200 + [4] * 200 + [2] * 200),   # ... forward workgroups loop too; lengths; In != H
    (33, 9, 28, 32, 4, None),                         # ... 32-wide variant, 4 layers
    (7, 6, 30, 30, 1, [6, 6, 5, 3, 3, 2, 1]),         # ... widths that are not multiples of 4, one layer
    (1100, 4, 24, 24, 3, None),                       # above 1024 sequences: MFMA whole-stack kernels
])
def test_dropin_module_matches_torch_cpu(B, T, In, H, L, lengths, packed):
    ref, ours = _pair(In, H, L)
    x = torch.rand(B, T, In, generator=torch.Generator().manual_seed(2))
    _compare_run(ref, ours, x, lengths, packed, f"B{B}T{T}In{In}H{H}L{L}{'p' if packed else ''}")


def test_same_module_called_three_times_in_one_graph():
    """d(h), d(h_hat), d(e_hat) share weights inside one graph (TimeGANv1.py:493-495): saved state
    must be per call."""
    ref, ours = _pair(24, 24, 3, seed=4)
    g = torch.Generator().manual_seed(3)
    xs = [torch.rand(8, 24, 24, generator=g) for _ in range(3)]
    lr = sum((ref(x)[0] ** 2).sum() for x in xs)
    lo = sum((ours(x.cuda())[0] ** 2).sum() for x in xs)
    lr.backward()
    lo.backward()
    for (name, pr_), (_, po_) in zip(ref.named_parameters(), ours.named_parameters()):
        assert_close(po_.grad.cpu().numpy(), pr_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, name)


def test_no_grad_and_time_major_and_hx():
    import timegan_amd as tg
    torch.manual_seed(6)
    ref = nn.GRU(12, 20, 2)                      # time-major, like metrics/GeneralRNN.py:41-44
    ours = tg.GRU(12, 20, 2).to("cuda:0")
    ours.load_state_dict(ref.state_dict())
    x = torch.rand(9, 5, 12)
    h0 = torch.randn(2, 5, 20)
    with torch.no_grad():
        yr, hr = ref(x, h0)
        yo, ho = ours(x.cuda(), h0.cuda())
    assert_close(yo.cpu().numpy(), yr.numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(ho.cpu().numpy(), hr.numpy(), FWD_RTOL, FWD_ATOL_SCALE, "h_n")
    # with gradients through hx
    xr = x.clone().requires_grad_(True); hr0 = h0.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True); ho0 = h0.clone().cuda().requires_grad_(True)
    (ref(xr, hr0)[0].sum() + ref(xr, hr0)[1].sum()).backward()
    (ours(xo, ho0)[0].sum() + ours(xo, ho0)[1].sum()).backward()
    assert_close(ho0.grad.cpu().numpy(), hr0.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dh0")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    for (name, pr_), (_, po_) in zip(ref.named_parameters(), ours.named_parameters()):
        assert_close(po_.grad.cpu().numpy(), pr_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, name)


def test_mid_hidden_with_initial_state_and_ragged_lengths():
    """H = 64 / 96 take the four-sequences-per-workgroup kernels and (B multiple of 16) the one-pass
    weight-gradient kernel, whose first rows pair with the initial state."""
    import timegan_amd as tg
    for (B, T, In, H, L, lens) in ((32, 6, 48, 64, 2, [6] * 10 + [4] * 12 + [1] * 10), (16, 5, 96, 96, 1, None)):
        torch.manual_seed(B + H)
        ref = nn.GRU(In, H, L, batch_first=True)
        ours = tg.GRU(In, H, L, batch_first=True).to("cuda:0")
        ours.load_state_dict(ref.state_dict())
        x = torch.rand(B, T, In)
        h0 = torch.randn(L, B, H)
        xr = x.clone().requires_grad_(True); hr0 = h0.clone().requires_grad_(True)
        xo = x.clone().cuda().requires_grad_(True); ho0 = h0.clone().cuda().requires_grad_(True)
        lens_ = [T] * B if lens is None else lens
        pr = nn.utils.rnn.pack_padded_sequence(xr, lens_, batch_first=True, enforce_sorted=True)
        yr, hnr = ref(pr, hr0)
        yr, _ = nn.utils.rnn.pad_packed_sequence(yr, batch_first=True, padding_value=0, total_length=T)
        yo, hno = ours(xo, ho0, lengths=lens)
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1))
        ((yr * gy).sum() + hnr.sum()).backward()
        ((yo * gy.cuda()).sum() + hno.sum()).backward()
        tag = f"H{H}"
        assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " y")
        assert_close(hno.detach().cpu().numpy(), hnr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " h_n")
        assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dx")
        assert_close(ho0.grad.cpu().numpy(), hr0.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dh0")
        for (name, p_), (_, q_) in zip(ref.named_parameters(), ours.named_parameters()):
            assert_close(q_.grad.cpu().numpy(), p_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " d" + name)


@pytest.mark.parametrize("switches", ["default", "no_fuse", "no_fuse_no_rows"])
@pytest.mark.parametrize("batch_first", [True, False])
def test_energy_width_with_initial_state_and_ragged_lengths(switches, batch_first, monkeypatch):
    """H = In = 96 has kernels of its own: the forward kernel that computes the input projection itself
    (TGRU_NO_FUSE=1 switches it off) and the row-streaming products for gi and dX (TGRU_NO_ROWS=1 sends those
    to the tile kernel).  All on the same case: 30 sequences (not a multiple of four: the last workgroup is
    short), ragged lengths, an initial state, both input layouts."""
    import timegan_amd as tg
    for k in ("TGRU_NO_FUSE", "TGRU_NO_ROWS"):
        monkeypatch.delenv(k, raising=False)
    if switches != "default":
        monkeypatch.setenv("TGRU_NO_FUSE", "1")
    if switches == "no_fuse_no_rows":
        monkeypatch.setenv("TGRU_NO_ROWS", "1")
    tg.reload_switches()
    B, T, H, L = 30, 7, 96, 2
    lens = [7] * 9 + [6] * 5 + [4] * 8 + [2] * 5 + [1] * 3
    torch.manual_seed(11)
    ref = nn.GRU(H, H, L, batch_first=batch_first)
    ours = tg.GRU(H, H, L, batch_first=batch_first).to("cuda:0")
    ours.load_state_dict(ref.state_dict())
    x = torch.rand(B, T, H) if batch_first else torch.rand(T, B, H)
    h0 = torch.randn(L, B, H)
    xr = x.clone().requires_grad_(True); hr0 = h0.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True); ho0 = h0.clone().cuda().requires_grad_(True)
    pr = nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=batch_first, enforce_sorted=True)
    yr, hnr = ref(pr, hr0)
    yr, _ = nn.utils.rnn.pad_packed_sequence(yr, batch_first=batch_first, padding_value=0, total_length=T)
    yo, hno = ours(xo, ho0, lengths=lens)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1))
    ((yr * gy).sum() + hnr.sum()).backward()
    ((yo * gy.cuda()).sum() + hno.sum()).backward()
    tag = f"H96 {switches} {'(B,T)' if batch_first else '(T,B)'}"
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " y")
    assert_close(hno.detach().cpu().numpy(), hnr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " h_n")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dx")
    assert_close(ho0.grad.cpu().numpy(), hr0.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dh0")
    for (name, p_), (_, q_) in zip(ref.named_parameters(), ours.named_parameters()):
        assert_close(q_.grad.cpu().numpy(), p_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " d" + name)


@pytest.mark.parametrize("family", ["persistent_tiles", "persistent_tiles_unfused", "persistent_tiles_staged",
                                    "persistent_tiles_streamed_backward", "persistent_tiles_wide_backward",
                                    "launch_per_step", "sixteen_sequences"])
def test_large_hidden_with_initial_state_and_ragged_lengths(family, monkeypatch):
    """H > 128 in multiples of 32 runs on 128-sequence x 32-unit tiles with the time loop inside the kernel
    (the workgroups of a row tile synchronise through counters; the forward kernel's operand tiles are fetched
    by loader waves and the layer's input projection runs inside it where the input width is a multiple of 32,
    TGRU_NO_STEP_FUSE=1 leaves the projection to the batched GEMM, TGRU_NO_STEP_DMA=1 stages the tiles through
    registers; the backward kernel works on 64 x 64
    tiles where H is a multiple of 64 -- with its tile of W_hh resident in registers and LDS, two chains of 32
    sequences worked alternately, at H = 256 / 384 / 512, TGRU_NO_STEP_RES=1 streams the W_hh tiles there too --
    and TGRU_STEP_BWD_WIDE=1 keeps it on 128 x 32); TGRU_NO_PERSIST=1 gives the
    same tiles with one launch per time step, TGRU_NO_STEP=1 the 16-sequence kernels: all on the same cases."""
    import timegan_amd as tg
    monkeypatch.delenv("TGRU_NO_STEP", raising=False)
    monkeypatch.delenv("TGRU_NO_PERSIST", raising=False)
    monkeypatch.delenv("TGRU_NO_STEP_DMA", raising=False)
    monkeypatch.delenv("TGRU_NO_STEP_FUSE", raising=False)
    monkeypatch.delenv("TGRU_STEP_BWD_WIDE", raising=False)
    monkeypatch.delenv("TGRU_NO_STEP_RES", raising=False)
    if family == "persistent_tiles_streamed_backward":
        monkeypatch.setenv("TGRU_NO_STEP_RES", "1")
    if family == "persistent_tiles_unfused":
        monkeypatch.setenv("TGRU_NO_STEP_FUSE", "1")
    elif family == "persistent_tiles_staged":
        monkeypatch.setenv("TGRU_NO_STEP_DMA", "1")
    elif family == "persistent_tiles_wide_backward":
        monkeypatch.setenv("TGRU_STEP_BWD_WIDE", "1")
    elif family == "launch_per_step":
        monkeypatch.setenv("TGRU_NO_PERSIST", "1")
    elif family == "sixteen_sequences":
        monkeypatch.setenv("TGRU_NO_STEP", "1")
    tg.reload_switches()
    for (B, T, In, H, L, lens) in ((300, 5, 40, 160, 2, [5] * 100 + [4] * 50 + [2] * 100 + [1] * 50),
                                   (130, 3, 256, 256, 1, None),
                                   (70, 4, 64, 512, 2, [4] * 30 + [3] * 20 + [1] * 20),
                                   (200, 3, 32, 384, 2, None),
                                   (5, 7, 20, 192, 3, [7, 7, 3, 2, 1])):
        torch.manual_seed(B + H)
        ref = nn.GRU(In, H, L, batch_first=True)
        ours = tg.GRU(In, H, L, batch_first=True).to("cuda:0")
        ours.load_state_dict(ref.state_dict())
        x = torch.rand(B, T, In)
        h0 = torch.randn(L, B, H)
        xr = x.clone().requires_grad_(True); hr0 = h0.clone().requires_grad_(True)
        xo = x.clone().cuda().requires_grad_(True); ho0 = h0.clone().cuda().requires_grad_(True)
        lens_ = [T] * B if lens is None else lens
        pr = nn.utils.rnn.pack_padded_sequence(xr, lens_, batch_first=True, enforce_sorted=True)
        yr, hnr = ref(pr, hr0)
        yr, _ = nn.utils.rnn.pad_packed_sequence(yr, batch_first=True, padding_value=0, total_length=T)
        yo, hno = ours(xo, ho0, lengths=lens)
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1))
        gh = torch.randn(hnr.shape, generator=torch.Generator().manual_seed(2))
        ((yr * gy).sum() + (hnr * gh).sum()).backward()
        ((yo * gy.cuda()).sum() + (hno * gh.cuda()).sum()).backward()
        tag = f"B{B}H{H} {family}"
        with torch.no_grad():   # inference: no reserve, the top layer's time-major copy lives in the workspace
            yi, _ = ours(x.cuda(), h0.cuda(), lengths=lens)
        assert_close(yi.cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " y (no_grad)")
        assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " y")
        assert_close(hno.detach().cpu().numpy(), hnr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, tag + " h_n")
        assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dx")
        assert_close(ho0.grad.cpu().numpy(), hr0.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " dh0")
        for (name, p_), (_, q_) in zip(ref.named_parameters(), ours.named_parameters()):
            assert_close(q_.grad.cpu().numpy(), p_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, tag + " d" + name)


def _raw_stack(tg, B, T, In, H, L, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / H ** 0.5
    ws = []
    for l in range(L):
        in_l = In if l == 0 else H
        for shape in ((3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)):
            ws.append(((torch.rand(shape, generator=g) * 2 - 1) * k).cuda())
    x = torch.rand(T, B, In, generator=g).cuda()
    dy = torch.randn(T, B, H, generator=g).cuda()
    out = tg.gru_forward_backward_raw(x, ws, B, T, tg.LAYOUT_TBF, dy)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("B,T,In,H,L,ragged,with_h0", [(300, 1, 64, 256, 2, False, False),     # one step: projection only
                                                        (300, 2, 32, 160, 3, True, False),     # one-tile projection
                                                        (130, 6, 512, 512, 2, True, True),
                                                        (257, 5, 96, 384, 2, False, True)])
def test_large_hidden_forward_with_the_projection_inside(B, T, In, H, L, ragged, with_h0, monkeypatch):
    """The persistent forward kernel that computes the layer's input projection itself (between a step's stores and
    the next step's recurrent product) against the same kernels fed by the batched projection GEMM: time-major
    rows through the C ABI, with and without an initial state, uniform and ragged lengths, a single step."""
    import timegan_amd as tg
    for k in ("TGRU_NO_STEP", "TGRU_NO_PERSIST", "TGRU_NO_STEP_DMA", "TGRU_NO_STEP_FUSE", "TGRU_STEP_BWD_WIDE"):
        monkeypatch.delenv(k, raising=False)
    g = torch.Generator().manual_seed(B + T)
    k = 1.0 / H ** 0.5
    ws = []
    for l in range(L):
        in_l = In if l == 0 else H
        for shape in ((3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)):
            ws.append(((torch.rand(shape, generator=g) * 2 - 1) * k).cuda())
    x = torch.rand(T, B, In, generator=g).cuda()
    dy = torch.randn(T, B, H, generator=g).cuda()
    h0 = torch.randn(L, B, H, generator=g).cuda() if with_h0 else None
    lens = None
    if ragged:
        lens = torch.tensor(sorted((1 + (7 * i) % T for i in range(B)), reverse=True), dtype=torch.int32).cuda()
    outs = []
    for fuse in (True, False):
        if not fuse:
            monkeypatch.setenv("TGRU_NO_STEP_FUSE", "1")
        tg.reload_switches()
        n0 = tg.launch_count()
        outs.append(tg.gru_forward_backward_raw(x, ws, B, T, tg.LAYOUT_TBF, dy, h0=h0, lengths=lens))
        outs[-1]["launches"] = tg.launch_count() - n0
        torch.cuda.synchronize()
    a, b = outs
    assert a["launches"] == b["launches"] - L          # one projection GEMM per layer less
    assert_close(a["y"].cpu().numpy(), b["y"].cpu().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(a["h_n"].cpu().numpy(), b["h_n"].cpu().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "h_n")
    assert_close(a["dx"].cpu().numpy(), b["dx"].cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    for i, (p, q) in enumerate(zip(a["grads"], b["grads"])):
        assert_close(p.cpu().numpy(), q.cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"grad {i}")
    assert tg.persist_timeouts() == 0


@pytest.mark.parametrize("B,T,In,H,L,ragged,with_h0", [(300, 6, 64, 512, 2, True, True),
                                                        (2048, 3, 512, 512, 1, False, False),    # one workgroup per CU
                                                        (2500, 4, 32, 256, 2, True, True),      # several launches
                                                        (129, 5, 96, 384, 3, False, True),
                                                        (3, 2, 32, 256, 1, False, True),        # second half of the tile empty
                                                        (70, 1, 64, 512, 2, False, True)])      # one step: dh0 only
def test_large_hidden_backward_with_resident_recurrent_weights(B, T, In, H, L, ragged, with_h0, monkeypatch):
    """The backward recurrence that keeps its tile of W_hh in registers and LDS (waves split k and fetch their own
    operand tiles, two chains of 32 sequences per workgroup with a counter each) against the kernel that streams
    the W_hh tiles, on the same calls: ragged lengths, an initial state (dh0 = one more product), batches that end
    inside a tile, more row tiles than one launch holds."""
    import timegan_amd as tg
    for k in ("TGRU_NO_STEP", "TGRU_NO_PERSIST", "TGRU_NO_STEP_DMA", "TGRU_NO_STEP_FUSE", "TGRU_STEP_BWD_WIDE",
              "TGRU_NO_STEP_RES"):
        monkeypatch.delenv(k, raising=False)
    g = torch.Generator().manual_seed(B + T)
    k = 1.0 / H ** 0.5
    ws = []
    for l in range(L):
        in_l = In if l == 0 else H
        for shape in ((3 * H, in_l), (3 * H, H), (3 * H,), (3 * H,)):
            ws.append(((torch.rand(shape, generator=g) * 2 - 1) * k).cuda())
    x = torch.rand(T, B, In, generator=g).cuda()
    dy = torch.randn(T, B, H, generator=g).cuda()
    dh_n = torch.randn(L, B, H, generator=g).cuda()
    h0 = torch.randn(L, B, H, generator=g).cuda() if with_h0 else None
    lens = None
    if ragged:
        lens = torch.tensor(sorted((1 + (7 * i) % T for i in range(B)), reverse=True), dtype=torch.int32).cuda()
    tg.persist_timeouts()
    outs = []
    for resident in (True, False):
        if not resident:
            monkeypatch.setenv("TGRU_NO_STEP_RES", "1")
        tg.reload_switches()
        outs.append(tg.gru_forward_backward_raw(x, ws, B, T, tg.LAYOUT_TBF, dy, h0=h0, dh_n=dh_n, lengths=lens))
        torch.cuda.synchronize()
    a, b = outs
    assert tg.persist_timeouts() == 0
    assert_close(a["dx"].cpu().numpy(), b["dx"].cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    if with_h0:
        assert_close(a["dh0"].cpu().numpy(), b["dh0"].cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dh0")
    for i, (p, q) in enumerate(zip(a["grads"], b["grads"])):
        assert_close(p.cpu().numpy(), q.cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"grad {i}")


def test_large_hidden_row_tiles_are_resident_per_xcd(monkeypatch):
    """A persistent launch must fit PER XCD (workgroup ids go round the eight XCDs, each XCD has an eighth of
    the compute units): 21 row tiles of twelve workgroups (B 2688, H 384) are three row tiles per XCD = 36
    workgroups for 32 compute units if the launch is sized by the chip's total.  The launcher splits such a
    batch into launches that fit; the result equals the one-launch-per-step kernels', and no wait timed out."""
    import timegan_amd as tg
    for k in ("TGRU_NO_STEP", "TGRU_NO_PERSIST", "TGRU_NO_STEP_DMA", "TGRU_STEP_BWD_WIDE", "TGRU_SPIN_LIMIT"):
        monkeypatch.delenv(k, raising=False)
    tg.reload_switches()
    tg.persist_timeouts()                                   # clear
    B, T, In, H, L = 2688, 3, 64, 384, 1
    a = _raw_stack(tg, B, T, In, H, L, 5)
    assert tg.persist_timeouts() == 0
    monkeypatch.setenv("TGRU_STEP_BWD_WIDE", "1")           # 128 x 32 backward tiles: twelve workgroups per row tile too
    tg.reload_switches()
    w = _raw_stack(tg, B, T, In, H, L, 5)
    assert tg.persist_timeouts() == 0
    monkeypatch.delenv("TGRU_STEP_BWD_WIDE")
    monkeypatch.setenv("TGRU_NO_PERSIST", "1")
    tg.reload_switches()
    b = _raw_stack(tg, B, T, In, H, L, 5)
    monkeypatch.delenv("TGRU_NO_PERSIST")
    tg.reload_switches()
    for got in (a, w):
        assert_close(got["y"].cpu().numpy(), b["y"].cpu().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
        assert_close(got["dx"].cpu().numpy(), b["dx"].cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
        for i, (p, q) in enumerate(zip(got["grads"], b["grads"])):
            assert_close(p.cpu().numpy(), q.cpu().numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"grad {i}")


def test_a_timed_out_wait_of_the_persistent_kernels_is_reported(monkeypatch):
    """TGRU_SPIN_LIMIT=0 makes every wait of the persistent H > 128 kernels give up at once, which is what a
    workgroup does whose peers never become resident.  The call must not pass for a good one: the heads of its
    results are NaN and the device's counter of such calls is non-zero; the next call with the default limit
    is clean again."""
    import timegan_amd as tg
    for k in ("TGRU_NO_STEP", "TGRU_NO_PERSIST", "TGRU_NO_STEP_DMA", "TGRU_STEP_BWD_WIDE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TGRU_SPIN_LIMIT", "0")
    tg.reload_switches()
    tg.persist_timeouts()
    out = _raw_stack(tg, 130, 4, 32, 256, 2, 7)
    assert torch.isnan(out["y"].reshape(-1)[:256]).all() and torch.isnan(out["h_n"].reshape(-1)[:256]).all()
    assert torch.isnan(out["dx"].reshape(-1)[:32]).all() and torch.isnan(out["grads"][0][0]).all()
    assert tg.persist_timeouts(clear=False) == 2            # the forward call and the backward call
    with pytest.raises(tg.TgruError, match="timed-out wait"):
        tg.check_persist_timeouts()
    monkeypatch.delenv("TGRU_SPIN_LIMIT")
    tg.reload_switches()
    out = _raw_stack(tg, 130, 4, 32, 256, 2, 7)
    assert tg.persist_timeouts() == 0
    assert not any(bool(torch.isnan(t).any()) for t in (out["y"], out["h_n"], out["dx"], *out["grads"]))


def test_reference_smoke_shapes_lengths_of_one_and_deep_stack():
    """run_embedding_test (TimeGANv1.py:77-94): x (10,150,5), t = ones(10) as a FLOAT tensor,
    H = 100, 50 layers -> (10,150,100); rows past length 1 are emb_linear(norm(0))."""
    import timegan_amd as tg
    cfg = tg.make_cfg(features=5, hidden=100, layers=50, seq_len=150, device="cuda:0")
    torch.manual_seed(8)
    cpu = tg.Embedding(cfg, rnn_cls=nn.GRU)
    gpu = tg.Embedding(cfg).to("cuda:0")
    gpu.load_state_dict(cpu.state_dict())
    x = torch.randn(10, 150, 5)
    t = torch.ones(10)
    with torch.no_grad():
        yr = cpu(x, t)
        yo = gpu(x.cuda(), t)
    assert tuple(yo.shape) == (10, 150, 100)
    assert_close(yo.cpu().numpy(), yr.numpy(), FWD_RTOL, 1e-5, "embedding smoke")
    pad_row = gpu.emb_linear(gpu.norm(torch.zeros(100, device="cuda:0")))
    assert torch.allclose(yo[3, 77], pad_row, atol=1e-6)


def test_unsorted_lengths_raise_like_torch():
    import timegan_amd as tg
    gru = tg.GRU(4, 8, 1, batch_first=True).to("cuda:0")
    with pytest.raises(RuntimeError, match="sorted in decreasing order"):
        gru(torch.zeros(3, 5, 4, device="cuda:0"), lengths=[2, 3, 5])
    with pytest.raises(RuntimeError, match="greater than 0"):
        gru(torch.zeros(3, 5, 4, device="cuda:0"), lengths=[3, 1, 0])


@pytest.mark.parametrize("dma", [True, False])
@pytest.mark.parametrize("B,T,In,H,L,time_major", [
    (24, 12, 160, 160, 2, False),     # 3H = 480: the last column tile hangs over the edge; batch-major rows
    (50, 7, 64, 224, 2, True),        # 350 rows: the last row tile hangs over; In != H
    (40, 33, 512, 512, 1, False),     # the stress width, row count not a multiple of the tile
])
def test_batched_products_with_and_without_the_loader_wave(B, T, In, H, L, time_major, dma, monkeypatch):
    """The batched products (input projection, its transpose, the split weight gradients) run on a kernel whose
    operand tiles are fetched by a loader wave with direct-to-LDS loads; TGRU_GEMM_NO_DMA=1 keeps them on the
    register-staged kernel.  Both against torch's CPU nn.GRU at shapes whose tiles hang over the edges."""
    import timegan_amd as tg
    if dma:
        monkeypatch.delenv("TGRU_GEMM_NO_DMA", raising=False)
    else:
        monkeypatch.setenv("TGRU_GEMM_NO_DMA", "1")
    tg.reload_switches()
    try:
        torch.manual_seed(11)
        ref = nn.GRU(In, H, L, batch_first=not time_major)
        ours = tg.GRU(In, H, L, batch_first=not time_major).to("cuda:0")
        ours.load_state_dict(ref.state_dict())
        shape = (T, B, In) if time_major else (B, T, In)
        x = torch.rand(shape, generator=torch.Generator().manual_seed(12))
        xr = x.clone().requires_grad_(True)
        xo = x.clone().cuda().requires_grad_(True)
        yr, hr = ref(xr)
        yo, ho = ours(xo)
        gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(13))
        (yr * gy).sum().backward()
        (yo * gy.cuda()).sum().backward()
        tag = f"B{B}T{T}In{In}H{H}L{L}{'dma' if dma else 'staged'}"
        assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, f"{tag} y")
        assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} dx")
        for (name, pr_), (_, po_) in zip(ref.named_parameters(), ours.named_parameters()):
            assert_close(po_.grad.cpu().numpy(), pr_.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"{tag} d{name}")
    finally:
        monkeypatch.delenv("TGRU_GEMM_NO_DMA", raising=False)
        tg.reload_switches()


@pytest.mark.parametrize("B,T,In,H,L", [(8, 6, 5, 12, 2),        # one sequence per workgroup
                                        (40, 6, 30, 30, 2),      # (packed below) MFMA whole-stack kernels
                                        (16, 6, 28, 96, 2),      # four-sequence kernels, dX inside at the upper layer
                                        (20, 8, 28, 40, 2),      # four-sequence kernels + row-split weight gradients
                                        (130, 4, 64, 160, 1)])   # per-tile kernels of the large hidden sizes
def test_frozen_weights_give_the_same_input_gradient(B, T, In, H, L):
    """A stack whose parameters want no gradient (a network the training step does not update): the library
    is handed NULL gradient arrays and launches no weight-gradient stage; dx and dh0 are bit for bit those of
    the full backward pass, and the parameters receive nothing."""
    import timegan_amd as tg
    torch.manual_seed(B + H)
    rnn = tg.GRU(In, H, num_layers=L, batch_first=True).to("cuda:0")
    x = torch.randn(B, T, In, device="cuda:0")
    h0 = torch.randn(L, B, H, device="cuda:0")
    dy = torch.randn(B, T, H, device="cuda:0")
    lengths = [T] * B if B != 40 else sorted(np.random.RandomState(0).randint(1, T + 1, B).tolist(), reverse=True)

    def run(frozen):
        for p in rnn.parameters():
            p.grad = None
            p.requires_grad_(not frozen)
        xi, hi = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        y, h_n = rnn(xi, hi, lengths=lengths)
        ((y * dy).sum() + h_n.sum()).backward()
        return xi.grad, hi.grad

    try:
        full = run(False)
        assert all(p.grad is not None for p in rnn.parameters())
        before = tg.lib().tgru_launch_count()
        part = run(True)
        launches_frozen = tg.lib().tgru_launch_count() - before
        before = tg.lib().tgru_launch_count()
        run(False)
        launches_full = tg.lib().tgru_launch_count() - before
    finally:
        for p in rnn.parameters():
            p.requires_grad_(True)
    assert all(p.grad is not None for p in rnn.parameters())      # from the last full run only
    assert torch.equal(part[0], full[0]) and torch.equal(part[1], full[1])
    assert launches_frozen < launches_full


@pytest.mark.parametrize("case", ["padded_lengths_h0", "time_major", "packed", "only_h_n", "frozen_weights", "no_grad"])
def test_cpp_autograd_binding_equals_the_ctypes_binding(case):
    """tg.GRU goes through the C++ autograd function of time-gan_amd/_tgru_torch.so by default (`cpp_op`); the same
    module with `cpp_op = False` goes through the ctypes binding.  Both call the same C ABI with the same buffers'
    contents, so outputs and gradients are bit-equal."""
    import timegan_amd as tg
    from timegan_amd import functional
    assert functional.cpp_op() is not None, "time-gan_amd/_tgru_torch.so is not built"
    torch.manual_seed(11)
    B, T, In, H, L = 9, 7, 5, 24, 2
    batch_first = case != "time_major"
    rnn = tg.GRU(In, H, L, batch_first=batch_first).to("cuda:0")
    if case == "frozen_weights":
        for p_ in rnn.parameters():
            p_.requires_grad_(False)
    lens = [7, 7, 6, 4, 4, 3, 2, 1, 1]
    x0 = torch.rand((B, T, In) if batch_first else (T, B, In), device="cuda:0")
    h00 = torch.randn(L, B, H, device="cuda:0")
    gy = torch.randn((B, T, H) if batch_first else (T, B, H), device="cuda:0")
    gh = torch.randn(L, B, H, device="cuda:0")
    results = []
    for cpp in (True, False):
        rnn.cpp_op = cpp
        rnn.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        h0 = h00.clone().requires_grad_(True)
        n0 = tg.launch_count()
        if case == "no_grad":
            with torch.no_grad():
                y, h_n = rnn(x, h0, lengths=lens)
            results.append([y, h_n])
            continue
        if case == "packed":
            packed = nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=True)
            yp, h_n = rnn(packed, h0)
            y, _ = nn.utils.rnn.pad_packed_sequence(yp, batch_first=True, total_length=T)
        elif case == "padded_lengths_h0" or case == "frozen_weights":
            y, h_n = rnn(x, h0, lengths=lens)
        else:
            y, h_n = rnn(x, h0)
        loss = (h_n * gh).sum() if case == "only_h_n" else (y * gy).sum() + (h_n * gh).sum()
        loss.backward()
        grads = [p_.grad for p_ in rnn.parameters()]
        if case == "frozen_weights":
            assert all(g is None for g in grads)
            grads = []
        results.append([y.detach(), h_n.detach(), x.grad, h0.grad, *grads])
        assert tg.launch_count() > n0
    a, b = results
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
