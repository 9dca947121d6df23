"""The loss builders of TimeGAN, mirrors of outdated_files/TimeGANv1.py:410-570.

Each function returns what its reference counterpart returns.  With one process they make the same
torch calls as the reference.  Under data parallelism (`group` with more than one rank) every mean
over the batch is rebuilt from all-reduced sums so that the value -- and, with gradients summed over
ranks, the gradient -- equals the single-process result on the concatenated batch:
  * sqrt(mse) terms (TimeGANv1.py:420-421,448-449,551): the mse is global before the sqrt;
  * the moment loss (TimeGANv1.py:543-548): batch mean and population std come from global sums;
  * BCE and plain MSE means: global sum / global count.
"""
from typing import Tuple

import torch
import torch.nn.functional as F

from . import loss_ops
from .modules import forward_pair
from .parallel import dist_sum, multi_rank


def _fused(kind, tensors, x_like, h_like, gamma=1.0, group=None):
    """The fused GPU op for loss `kind` (loss_ops) on (B,T,F)-shaped `x_like` / (B,T,H)-shaped `h_like`
    operands, or None when the tensors are not fp32 GPU tensors (the CPU path of the tests and of the
    CPU baseline keeps the literal torch composition below)."""
    if not loss_ops.usable(*tensors):
        return None
    ref = h_like if h_like is not None else x_like
    B, T = ref.shape[0], ref.shape[1]
    H = h_like.shape[2] if h_like is not None else x_like.shape[2]
    F = x_like.shape[2] if x_like is not None else h_like.shape[2]
    return loss_ops.fused_loss(kind, tensors, B, T, H, F, gamma, group)


def _count(t: torch.Tensor, group) -> torch.Tensor:
    n = torch.tensor(float(t.numel()), device=t.device)
    return dist_sum(n, group) if multi_rank(group) else n


def _mse(a: torch.Tensor, b: torch.Tensor, group=None) -> torch.Tensor:
    if not multi_rank(group):
        return F.mse_loss(a, b)
    return dist_sum(((a - b) ** 2).sum(), group) / _count(a, group)


def _bce_logits(y: torch.Tensor, target_one: bool, group=None) -> torch.Tensor:
    target = torch.ones_like(y) if target_one else torch.zeros_like(y)
    if not multi_rank(group):
        return F.binary_cross_entropy_with_logits(y, target)
    s = F.binary_cross_entropy_with_logits(y, target, reduction="sum")
    return dist_sum(s, group) / _count(y, group)


def _batch_mean_std(v: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mean(dim=0) and std(dim=0, unbiased=False) over the GLOBAL batch."""
    if not multi_rank(group):
        return v.mean(dim=0), v.std(dim=0, unbiased=False)
    # Everything computed after a dist_sum must be the same on every rank (its backward is the
    # identity), so the variance comes from all-reduced raw moments, taken about a constant close to
    # the mean (the detached global mean) to keep E[d^2] - E[d]^2 free of cancellation.
    n = dist_sum(torch.tensor(float(v.shape[0]), device=v.device), group)
    c = (dist_sum(v.detach().sum(dim=0), group) / n).detach()
    dlt = v - c
    m1 = dist_sum(dlt.sum(dim=0), group) / n
    m2 = dist_sum((dlt * dlt).sum(dim=0), group) / n
    return c + m1, torch.sqrt(torch.clamp(m2 - m1 * m1, min=0.0))


# The H > 128 kernels address a batch-major (B, T, H) tensor through 32-bit byte offsets; a batched call that reaches
# 2 GiB would leave their persistent path (one launch per time step instead), so such calls stay separate.
_BATCHED_BYTES_LIMIT = 1 << 31


def _batched(inputs, t, hidden=0):
    """The inputs of several same-weight calls as ONE batch, or None when lengths are ragged (a
    concatenation of sorted length lists is not sorted) or when the batch -- rows x T x the wider of the input and
    the network's hidden size `hidden` -- would reach 2 GiB: (concatenated input, its lengths, the sizes)."""
    T = inputs[0].shape[1]
    lens = [int(v) for v in t]
    if len(inputs) == 1 or any(v != T for v in lens):
        return None
    rows = sum(u.shape[0] for u in inputs)
    if rows * T * max(int(inputs[0].shape[2]), int(hidden)) * 4 >= _BATCHED_BYTES_LIMIT:
        return None
    return torch.cat(inputs, dim=0), lens * len(inputs), [u.shape[0] for u in inputs]


def _apply_batched(net, inputs, t):
    """net(u, t) for each u in `inputs`, evaluated as ONE call on the concatenated batch when every
    sequence has full length (rows of a batch never interact inside a network, so the results are
    those of the separate calls the reference makes, e.g. d(h), d(_h), d(_e) at TimeGANv1.py:493-495;
    the weight gradients come out already summed).  Ragged lengths, and batches of 2 GiB, keep separate calls."""
    both = _batched(inputs, t, getattr(net, "dim_hidden", 0))
    if both is None:
        return [net(u, t) for u in inputs]
    return net(both[0], both[1]).split(both[2], dim=0)


def embedding_forward_side(emb, rec, x, t, group=None):
    """TimeGANv1.py:410-423 -> (e_loss0, x_tilde)."""
    h = emb(x, t)
    x_tilde = rec(h, t)
    out = _fused(loss_ops.EMBED_SIDE, [x_tilde, x], x, None, group=group)
    if out is not None:
        return out[0], x_tilde
    e_loss_t0 = _mse(x_tilde, x, group)
    return 10 * torch.sqrt(e_loss_t0), x_tilde


def embedding_forward_main(emb, sup, rec, x, t, group=None, h=None):
    """TimeGANv1.py:426-451 -> (e_loss, e_loss0, e_loss_t0).  `h`: emb(x, t) evaluated by the caller with these
    same weights (it stays in the autograd graph: the embedder's gradients flow through it)."""
    if h is None:
        h = emb(x, t)
    x_tilde, h_sup = forward_pair(rec, h, t, sup, h, t)
    out = _fused(loss_ops.EMBED_MAIN, [x_tilde, x, h_sup, h], x, h, group=group)
    if out is not None:
        return out
    g_loss_sup = _mse(h_sup[:, :-1, :], h[:, 1:, :], group)
    e_loss_t0 = _mse(x_tilde, x, group)
    e_loss0 = 10 * torch.sqrt(e_loss_t0)
    return e_loss0 + 0.1 * g_loss_sup, e_loss0, e_loss_t0


def supervisor_forward(emb, sup, x, t, group=None):
    """TimeGANv1.py:454-472 -> (s_loss, h, h_sup)."""
    h = emb(x, t)
    h_sup = sup(h, t)
    out = _fused(loss_ops.SUPERVISOR, [h_sup, h], None, h, group=group)
    if out is not None:
        return out[0], h, h_sup
    return _mse(h_sup[:, :-1, :], h[:, 1:, :], group), h, h_sup


def discriminator_forward(emb, sup, g, d, x, t, z, gamma=1.0, group=None):
    """TimeGANv1.py:475-503 -> d_loss.  Only the discriminator receives gradients."""
    with torch.no_grad():   # the reference detaches all three (TimeGANv1.py:486-491): nothing to save for backward
        h, e_hat = forward_pair(emb, x, t, g, z, t)
        h_hat = sup(h, t)
    y_real, y_fake, y_fake_e = _apply_batched(d, [h, h_hat, e_hat], t)
    out = _fused(loss_ops.DISCRIMINATOR, [y_real, y_fake, y_fake_e], None, h, gamma, group)
    if out is not None:
        return out[0]
    d_loss_real = _bce_logits(y_real, True, group)
    d_loss_fake = _bce_logits(y_fake, False, group)
    d_loss_fake_e = _bce_logits(y_fake_e, False, group)
    return d_loss_fake + d_loss_real + gamma * d_loss_fake_e


def generator_forward(emb, sup, g, d, rec, x, t, z, gamma=1.0, group=None, h=None):
    """TimeGANv1.py:506-553 -> g_loss.  `h`: the VALUE of emb(x, t) (detached) when the caller has it already and
    the embedder's gradients are not wanted from this loss; the embedder is not evaluated then.

    The reference also evaluates rec(h) at :521 and overwrites the result at :528 before any use;
    that forward contributes nothing to the loss or to any gradient and is not evaluated here.
    """
    if h is None:
        h, e_hat = forward_pair(emb, x, t, g, z, t)
    else:
        e_hat = g(z, t)
    h_sup, h_hat = _apply_batched(sup, [h, e_hat], t)
    both = _batched([h_hat, e_hat], t, getattr(d, "dim_hidden", 0))
    if both is None:
        x_hat = rec(h_hat, t)
        y_fake, y_fake_e = d(h_hat, t), d(e_hat, t)
    else:
        x_hat, y = forward_pair(rec, h_hat, t, d, both[0], both[1])
        y_fake, y_fake_e = y.split(both[2], dim=0)
    out = _fused(loss_ops.GENERATOR, [y_fake, y_fake_e, h_sup, h, x_hat, x], x, h, gamma, group)
    if out is not None:
        return out[0]
    g_loss_u = _bce_logits(y_fake, True, group)
    g_loss_u_e = _bce_logits(y_fake_e, True, group)
    g_loss_s = _mse(h_sup[:, :-1, :], h[:, 1:, :], group)
    m_hat, s_hat = _batch_mean_std(x_hat, group)
    m_x, s_x = _batch_mean_std(x, group)
    # precedence exactly as written at TimeGANv1.py:543-544
    g_loss_v1 = torch.mean(torch.abs((s_hat + 1e-6) - s_x + 1e-6))
    g_loss_v2 = torch.mean(torch.abs(m_hat - m_x))
    return g_loss_u + gamma * g_loss_u_e + 100 * torch.sqrt(g_loss_s) + 1000 * (g_loss_v1 + g_loss_v2)


def inference(sup, g, rec, z, t):
    """TimeGANv1.py:556-570: noise -> synthetic sequences."""
    return rec(sup(g(z, t), t), t)
