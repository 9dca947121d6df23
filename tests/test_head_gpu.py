"""GPU tests of the fused dense head (tgru_head_forward / tgru_head_backward) against the torch
modules it replaces (nn.LayerNorm / nn.Linear / ReLU evaluated on the CPU in fp32), for every head
shape the five TimeGAN networks use and for ragged sizes."""
import copy

import pytest
import torch
import torch.nn as nn

from helpers import FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close

pytestmark = pytest.mark.gpu

# (rows, din, dmid, dout, input LayerNorm, deep)
CASES = [
    (128 * 24, 6, 24, 24, False, True),     # embedder feature MLP (stock: 6 features)
    (128 * 24, 24, 24, 24, True, False),    # norm + emb_linear / g_linear
    (128 * 24, 24, 24, 6, True, True),      # recovery: norm + mlp(H -> H -> F)
    (256 * 24, 24, 24, 24, True, True),     # supervisor on a doubled batch
    (384 * 24, 24, 24, 1, True, False),     # discriminator: norm + Linear(H, 1)
    (37, 5, 10, 7, True, True),             # ragged rows, widths that are not multiples of 4
    (1, 32, 32, 32, True, True),            # a single row at the maximum width
    (100, 28, 32, 32, False, False),
    (128 * 24, 6, 48, 24, False, True),     # generator feature MLP: F -> 2H -> H
    (53, 30, 64, 31, True, True),           # the widest middle, odd widths around it
    (40, 16, 33, 16, True, True),           # a middle just past two tiles
    (16, 8, 16, 4, False, True),            # one tile everywhere
    # beyond the register-resident kernels: weights in LDS, weight gradients as row-split products
    (1024 * 24, 96, 96, 96, True, True),    # supervisor at dim_hidden 96, batch 1024
    (1024 * 24, 96, 96, 28, True, True),    # recovery: norm + mlp(H -> H -> F)
    (1024 * 24, 28, 96, 96, False, True),   # embedder feature MLP
    (1024 * 24, 28, 192, 96, False, True),  # generator feature MLP: F -> 2H -> H
    (1024 * 24, 96, 96, 96, True, False),   # norm + emb_linear / g_linear
    (2048 * 24, 96, 1, 1, True, False),     # discriminator on a doubled batch: norm + Linear(H, 1)
    (37, 70, 130, 45, True, True),          # ragged rows, odd widths in every position
    (100, 128, 96, 128, True, True),        # the widest input / output
    (53, 100, 128, 128, True, False),       # a wide head without the second Linear
    (1, 96, 96, 96, True, True),            # a single row
    (3000, 33, 65, 33, False, True),        # just past the narrow kernels' limits
]


def _modules(din, dmid, dout, ln0, deep, seed):
    torch.manual_seed(seed)
    m = {"ln0": nn.LayerNorm(din) if ln0 else None, "lin1": nn.Linear(din, dmid),
         "ln1": nn.LayerNorm(dmid) if deep else None, "lin2": nn.Linear(dmid, dout) if deep else None}
    for k in ("ln0", "ln1"):   # non-trivial affine parameters
        if m[k] is not None:
            with torch.no_grad():
                m[k].weight.uniform_(0.5, 1.5)
                m[k].bias.uniform_(-0.5, 0.5)
    return m


def _torch_head(m, x):
    h = m["lin1"](x if m["ln0"] is None else m["ln0"](x))
    return h if m["lin2"] is None else m["lin2"](torch.relu(m["ln1"](h)))


def _off_the_kink(m, x, gen, margin=1e-4):
    """Rows whose pre-activation lies within `margin` of the ReLU kink are redrawn: two correct fp32
    implementations may take different branches there, and with millions of pre-activations some do."""
    if m["lin2"] is None:
        return x
    with torch.no_grad():
        for _ in range(20):
            pre = m["ln1"](m["lin1"](x if m["ln0"] is None else m["ln0"](x)))
            near = (pre.abs() < margin).any(-1)
            if not near.any():
                break
            x[near] = torch.randn(int(near.sum()), x.shape[1], generator=gen)
    return x


@pytest.mark.parametrize("case", CASES, ids=lambda c: "m%d_%d_%d_%d_ln%d_deep%d" % c)
@pytest.mark.parametrize("want_dx", [True, False])
def test_head_matches_torch(case, want_dx):
    import copy
    from timegan_amd.head import head_apply, supported
    M, din, dmid, dout, ln0, deep = case
    ref = _modules(din, dmid, dout, ln0, deep, seed=M + din)
    ours = {k: (None if v is None else copy.deepcopy(v).to("cuda:0")) for k, v in ref.items()}
    g = torch.Generator().manual_seed(3)
    x = _off_the_kink(ref, torch.randn(M, din, generator=g), g)
    dy = torch.randn(M, dout, generator=g)
    xr = x.clone().requires_grad_(want_dx)
    xo = x.clone().to("cuda:0").requires_grad_(want_dx)
    assert supported(xo, ours["lin1"], ours["lin2"])
    yr = _torch_head(ref, xr)
    yo = head_apply(xo, ours["ln0"], ours["lin1"], ours["ln1"], ours["lin2"])
    (yr * dy).sum().backward()
    (yo * dy.cuda()).sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    if want_dx:
        assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    for k in ref:
        if ref[k] is None:
            continue
        for (name, pr), (_, po) in zip(ref[k].named_parameters(), ours[k].named_parameters()):
            assert_close(po.grad.cpu().numpy(), pr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"d{k}.{name}")


def test_head_3d_input_and_wide_fallback():
    """(B, T, F) inputs keep their leading shape; widths beyond the kernels' limits (128 in / out,
    192 in the middle) run the torch modules on the GPU."""
    from timegan_amd.head import head_apply, supported
    m = _modules(6, 200, 24, False, True, seed=0)
    m = {k: (None if v is None else v.to("cuda:0")) for k, v in m.items()}
    x = torch.rand(4, 5, 6, device="cuda:0")
    assert not supported(x, m["lin1"], m["lin2"])
    y = head_apply(x, None, m["lin1"], m["ln1"], m["lin2"])
    assert y.shape == (4, 5, 24)
    m2 = _modules(6, 24, 24, False, True, seed=0)
    m2 = {k: (None if v is None else v.to("cuda:0")) for k, v in m2.items()}
    y2 = head_apply(x, None, m2["lin1"], m2["ln1"], m2["lin2"])
    ref = _torch_head({k: (None if v is None else v.cpu()) for k, v in m2.items()}, x.cpu())
    assert y2.shape == (4, 5, 24)
    assert_close(y2.detach().cpu().numpy(), ref.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y 3d")


@pytest.mark.parametrize("rows,din,dmid,dout", [(4096, 96, 96, 96), (3000, 28, 96, 96), (2500, 30, 70, 45)])
def test_wide_head_weight_gradient_split_over_rows(rows, din, dmid, dout, monkeypatch):
    """Heads the fused kernels do not take (here: switched off): torch's modules, with the Linear weight
    gradients (out x in products over all rows) formed by the library's row-split GEMM and LayerNorm on the
    library's kernels; everything against torch on the CPU."""
    import timegan_amd as tg
    from timegan_amd.head import WIDE_LINEAR_MIN_ROWS, head_apply, supported
    monkeypatch.setenv("TGRU_NO_WIDE_HEAD", "1")
    tg.reload_switches()
    assert rows >= WIDE_LINEAR_MIN_ROWS
    ref = _modules(din, dmid, dout, True, True, seed=3)
    ours = {k: (None if v is None else copy.deepcopy(v).to("cuda:0")) for k, v in ref.items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, din, generator=g)
    dy = torch.randn(rows, dout, generator=g)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True)
    assert not supported(xo, ours["lin1"], ours["lin2"])
    yr = _torch_head(ref, xr)
    yo = head_apply(xo, ours["ln0"], ours["lin1"], ours["ln1"], ours["lin2"])
    (yr * dy).sum().backward()
    (yo * dy.cuda()).sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    for k in ref:
        if ref[k] is None:
            continue
        for (name, pr), (_, po) in zip(ref[k].named_parameters(), ours[k].named_parameters()):
            assert_close(po.grad.cpu().numpy(), pr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"d{k}.{name}")


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("rows,width", [(2304, 512), (2051, 1024), (4100, 160), (2048, 768), (3000, 96), (2500, 132)])
def test_layer_norm_up_to_1024_wide_with_relu(rows, width, relu):
    """The wave-per-row LayerNorm kernels (widths above 128, and any width with the ReLU in the same pass):
    relu(LayerNorm(x)) and its gradients against torch on the CPU, the stress configuration's widths included
    (512; 1024 inside the generator's feature MLP)."""
    from timegan_amd.head import wide_layer_norm
    torch.manual_seed(rows + width)
    ref = nn.LayerNorm(width)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    ours = copy.deepcopy(ref).to("cuda:0")
    x = torch.randn(rows, width) * 2 + 0.5
    dy = torch.randn(rows, width)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True)
    yr = torch.relu(ref(xr)) if relu else ref(xr)
    yo = wide_layer_norm(xo, ours, relu=relu)
    (yr * dy).sum().backward()
    (yo * dy.cuda()).sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    assert_close(ours.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dgamma")
    assert_close(ours.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dbeta")


@pytest.mark.parametrize("rows,din,dmid,dout,ln0", [(2304, 64, 512, 512, False),     # embedder's feature MLP at H 512
                                                     (2304, 64, 1024, 512, False),    # generator's
                                                     (2304, 512, 512, 64, True),      # recovery's output stage
                                                     (2304, 512, 512, 512, True),     # supervisor's
                                                     (2304, 160, 160, 160, True)])
def test_heads_at_the_stress_widths_run_on_the_library(rows, din, dmid, dout, ln0):
    """Heads beyond the fused kernels' widths (hidden size 512): every product on the library's GEMMs, LayerNorm
    (+ ReLU) on its wave-per-row kernels, bias gradients as column sums -- no vendor GEMM and no torch
    LayerNorm launch -- against torch on the CPU at the standard bar."""
    import timegan_amd as tg
    from timegan_amd.head import head_apply, supported
    ref = _modules(din, dmid, dout, ln0, True, seed=5)
    ours = {k: (None if v is None else copy.deepcopy(v).to("cuda:0")) for k, v in ref.items()}
    g = torch.Generator().manual_seed(11)
    x = _off_the_kink(ref, torch.rand(rows, din, generator=g), g)
    dy = torch.randn(rows, dout, generator=g)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True)
    assert not supported(xo, ours["lin1"], ours["lin2"])
    yr = _torch_head(ref, xr)
    n0 = tg.launch_count()
    yo = head_apply(xo, ours["ln0"], ours["lin1"], ours["ln1"], ours["lin2"])
    assert tg.launch_count() - n0 == (4 if ln0 else 3)       # [LN0], GEMM, LN1 + ReLU, GEMM
    (yr * dy).sum().backward()
    (yo * dy.cuda()).sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    for k in ref:
        if ref[k] is None:
            continue
        for (name, pr), (_, po) in zip(ref[k].named_parameters(), ours[k].named_parameters()):
            assert_close(po.grad.cpu().numpy(), pr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, f"d{k}.{name}")


@pytest.mark.parametrize("rows,width,affine", [(4096, 96, True), (2051, 28, True), (3000, 128, True), (2500, 4, True),
                                                 (2048, 64, False)])
def test_wide_layer_norm_matches_torch_cpu(rows, width, affine):
    """tgru_layer_norm_forward / backward behind nn.LayerNorm's parameters, against torch on the CPU."""
    from timegan_amd.head import wide_layer_norm
    torch.manual_seed(rows + width)
    ref = nn.LayerNorm(width, elementwise_affine=affine)
    if affine:
        with torch.no_grad():
            ref.weight.uniform_(0.5, 1.5)
            ref.bias.uniform_(-0.5, 0.5)
    ours = copy.deepcopy(ref).to("cuda:0")
    x = torch.randn(rows // 8, 8, width) * 2 + 0.5       # 3-D input: the leading shape must survive
    dy = torch.randn(rows // 8, 8, width)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().cuda().requires_grad_(True)
    yr = ref(xr)
    yo = wide_layer_norm(xo, ours)
    assert yo.shape == yr.shape
    (yr * dy).sum().backward()
    (yo * dy.cuda()).sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "y")
    assert_close(xo.grad.cpu().numpy(), xr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dx")
    if affine:
        assert_close(ours.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dgamma")
        assert_close(ours.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, "dbeta")
    # widths the kernels do not take fall back to torch
    odd = nn.LayerNorm(30).to("cuda:0")
    assert wide_layer_norm(torch.randn(4096, 30, device="cuda:0"), odd).shape == (4096, 30)


PAIRS = [
    ((3072, 24, 24, 24, True, False), (3072, 24, 24, 24, True, False)),   # embedder + generator output stages
    ((3072, 24, 24, 6, True, True), (6144, 24, 24, 1, True, False)),      # recovery + discriminator (doubled batch)
    ((3072, 6, 24, 24, False, True), (3072, 6, 48, 24, False, True)),     # the two feature MLPs, one of them wide
    ((37, 5, 10, 7, True, True), (20, 30, 64, 31, True, True)),
    ((3072, 96, 96, 96, True, True), (3072, 24, 24, 24, True, False)),    # one of them beyond the paired kernels
    ((3072, 96, 96, 28, True, True), (6144, 96, 1, 1, True, False)),      # both (issued one after the other)
]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: "%d_%d_%d_%d__%d_%d_%d_%d" % (p[0][:4] + p[1][:4]))
def test_head_pair_equals_separate_calls(pair):
    """Two heads in one launch give bit-for-bit what the two separate launches give."""
    from timegan_amd.head import head_apply, head_apply_pair
    mods, xs, dys = [], [], []
    for k, (M, din, dmid, dout, ln0, deep) in enumerate(pair):
        m = _modules(din, dmid, dout, ln0, deep, seed=17 + k)
        mods.append({n: (None if v is None else v.to("cuda:0")) for n, v in m.items()})
        g = torch.Generator().manual_seed(5 + k)
        xs.append(torch.randn(M, din, generator=g).to("cuda:0"))
        dys.append(torch.randn(M, dout, generator=g).to("cuda:0"))
    spec = [(m["ln0"], m["lin1"], m["ln1"], m["lin2"]) for m in mods]

    def run(paired):
        for m in mods:
            for v in m.values():
                if v is not None:
                    v.zero_grad()
        xa, xb = (x.clone().requires_grad_(True) for x in xs)
        if paired:
            ya, yb = head_apply_pair(xa, spec[0], xb, spec[1])
        else:
            ya, yb = head_apply(xa, *spec[0]), head_apply(xb, *spec[1])
        ((ya * dys[0]).sum() + (yb * dys[1]).sum()).backward()
        grads = [p.grad.clone() for m in mods for v in m.values() if v is not None for p in v.parameters()]
        return [ya.detach(), yb.detach(), xa.grad, xb.grad] + grads

    ref, got = run(False), run(True)
    for i, (u, v) in enumerate(zip(got, ref)):
        assert torch.equal(u, v), f"tensor {i}: max diff {float((u - v).abs().max()):.3e}"


@pytest.mark.parametrize("case", [(3072, 24, 24, 6, True, True), (3072, 6, 48, 24, False, True), (3072, 24, 24, 24, True, False),
                                  (4096, 96, 96, 28, True, True), (4096, 28, 192, 96, False, True),
                                  (4096, 96, 96, 96, True, False)], ids=lambda c: "m%d_%d_%d_%d_ln%d_deep%d" % c)
def test_frozen_head_gives_the_same_input_gradient(case):
    """A head whose parameters want no gradient: dparams NULL, no reduction / weight-gradient products, dx bit
    for bit that of the full backward pass."""
    import timegan_amd as tg
    from timegan_amd.head import head_apply
    M, din, dmid, dout, ln0, deep = case
    m = {k: (None if v is None else v.to("cuda:0")) for k, v in _modules(din, dmid, dout, ln0, deep, seed=11).items()}
    params = [p for v in m.values() if v is not None for p in v.parameters()]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, din, generator=g).cuda()
    dy = torch.randn(M, dout, generator=g).cuda()

    def run(frozen):
        for p in params:
            p.grad = None
            p.requires_grad_(not frozen)
        xi = x.clone().requires_grad_(True)
        before = tg.lib().tgru_launch_count()
        (head_apply(xi, m["ln0"], m["lin1"], m["ln1"], m["lin2"]) * dy).sum().backward()
        return xi.grad, tg.lib().tgru_launch_count() - before

    full, n_full = run(False)
    part, n_part = run(True)
    assert all(p.grad is None for p in params)
    assert torch.equal(part, full)
    assert n_part < n_full
