"""The five TimeGAN networks, host-side mirrors of outdated_files/TimeGANv1.py:17-388.

Same constructor contract (a nested cfg dict, config/config.yaml:1-38), same sub-module names and
creation order (so state_dict keys, shapes and same-seed initial values match the reference and its
`.pt` files interchange), same forward signature `forward(x, t)` with `t` the per-sequence lengths.

Differences, all output-preserving:
  * the recurrent layer is `timegan_amd.GRU` (HIP kernels) instead of torch.nn.GRU;
  * the dense stages around it (LayerNorm / Linear / ReLU chains) run as one fused kernel each
    (`head.py`) when every width is <= 32, through the very same torch parameters;
  * pack_padded_sequence / pad_packed_sequence (TimeGANv1.py:56-65) are not materialised: lengths go
    to the kernels, which skip finished sequences and write the padding rows themselves.

`rnn_cls` exists so that the CPU tests and the CPU baseline can build the same networks on
torch.nn.GRU; with it the forward takes the reference's literal pack -> GRU -> pad route.
"""
from typing import Dict

import torch
import torch.nn as nn

from .gru import GRU, gru_pair
from .head import head_apply, head_apply_pair


class _RecurrentBlock(nn.Module):
    """Shared forward plumbing: pre (the feature MLP, if any) -> GRU over valid steps -> post (LayerNorm
    and the output layers).  Subclasses define `pre` / `post` and name their GRU in `_rnn_name`."""

    def pre_spec(self):
        """(ln0, lin1, ln1, lin2) of the stage before the GRU, or None when there is none."""
        return None

    def post_spec(self):
        """(ln0, lin1, ln1, lin2) of the stage after the GRU."""
        raise NotImplementedError

    def _stage(self, x: torch.Tensor, spec) -> torch.Tensor:
        if spec is None:
            return x
        return self._head(x, *spec)

    def pre(self, x: torch.Tensor) -> torch.Tensor:
        return self._stage(x, self.pre_spec())

    def post(self, h: torch.Tensor) -> torch.Tensor:
        return self._stage(h, self.post_spec())

    def forward(self, x: torch.Tensor, t) -> torch.Tensor:
        return self.post(self._run_rnn(getattr(self, self._rnn_name), self.pre(x), t))

    def _run_rnn(self, rnn: nn.Module, x: torch.Tensor, t) -> torch.Tensor:
        if isinstance(rnn, GRU):
            rnn.cpp_op = False      # (the trainer sums these gradients through the deferred-reduction queue)
            h, _ = rnn(x, lengths=t)
            if self.padding_value != 0:
                lens = torch.as_tensor([int(v) for v in t], device=x.device)
                pad = torch.arange(self.seq_len, device=x.device)[None, :] >= lens[:, None]
                h = h.masked_fill(pad[:, :, None], float(self.padding_value))
            return h
        packed = nn.utils.rnn.pack_padded_sequence(input=x, lengths=t, batch_first=True, enforce_sorted=True)
        h, _ = rnn(packed)
        h, _ = nn.utils.rnn.pad_packed_sequence(sequence=h, batch_first=True,
                                                padding_value=self.padding_value, total_length=self.seq_len)
        return h

    def _head(self, x: torch.Tensor, ln0, lin1, ln1=None, lin2=None) -> torch.Tensor:
        """[ln0] -> lin1 -> [ln1 -> ReLU -> lin2]: one fused kernel on the GPU path, the torch modules
        themselves on the CPU path (`rnn_cls=torch.nn.GRU`)."""
        if x.is_cuda and isinstance(getattr(self, self._rnn_name), GRU):
            return head_apply(x, ln0, lin1, ln1, lin2)
        h = lin1(x if ln0 is None else ln0(x))
        return h if lin2 is None else lin2(torch.relu(ln1(h)))

    @property
    def device(self):
        return next(self.parameters()).device


class Embedding(_RecurrentBlock):
    """Encoder x -> H (TimeGANv1.py:17-74): mlp(F->H, LN, ReLU, H->H) -> GRU -> LN -> Linear."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        self.dim_features = int(cfg['emb']['dim_features'])
        self.dim_hidden = int(cfg['emb']['dim_hidden'])
        self.num_layers = int(cfg['emb']['num_layers'])
        self.seq_len = int(cfg['system']['seq_len'])
        self.padding_value = int(cfg['system']['padding_value'])
        self.emb_rnn = rnn_cls(input_size=self.dim_hidden, hidden_size=self.dim_hidden,
                               num_layers=self.num_layers, batch_first=True)
        self.emb_linear = nn.Linear(self.dim_hidden, self.dim_hidden)
        self.norm = nn.LayerNorm(self.dim_hidden)
        self.mlp = nn.Sequential(nn.Linear(self.dim_features, self.dim_hidden), nn.LayerNorm(self.dim_hidden),
                                 nn.ReLU(), nn.Linear(self.dim_hidden, self.dim_hidden))

    _rnn_name = "emb_rnn"

    def pre_spec(self):
        return (None, self.mlp[0], self.mlp[1], self.mlp[3])

    def post_spec(self):
        return (self.norm, self.emb_linear, None, None)


class Recovery(_RecurrentBlock):
    """Decoder H -> x (TimeGANv1.py:97-153): GRU -> LN -> mlp(H->H, LN, ReLU, H->F)."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        self.dim_output = int(cfg['rec']['dim_output'])
        self.dim_hidden = int(cfg['rec']['dim_hidden'])
        self.num_layers = int(cfg['rec']['num_layers'])
        self.seq_len = int(cfg['system']['seq_len'])
        self.padding_value = int(cfg['system']['padding_value'])
        self.rec_rnn = rnn_cls(input_size=self.dim_hidden, hidden_size=self.dim_hidden,
                               num_layers=self.num_layers, batch_first=True)
        self.mlp = nn.Sequential(nn.Linear(self.dim_hidden, self.dim_hidden), nn.LayerNorm(self.dim_hidden),
                                 nn.ReLU(), nn.Linear(self.dim_hidden, self.dim_output))
        self.norm = nn.LayerNorm(self.dim_hidden)

    _rnn_name = "rec_rnn"

    def post_spec(self):
        return (self.norm, self.mlp[0], self.mlp[1], self.mlp[3])


class Supervisor(_RecurrentBlock):
    """Next-step predictor in latent space (TimeGANv1.py:176-232): GRU -> LN -> mlp(H->H, LN, ReLU, H->H)."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        self.dim_hidden = int(cfg['sup']['dim_hidden'])
        self.num_layers = int(cfg['sup']['num_layers'])
        self.seq_len = int(cfg['system']['seq_len'])
        self.padding_value = int(cfg['system']['padding_value'])
        self.sup_rnn = rnn_cls(input_size=self.dim_hidden, hidden_size=self.dim_hidden,
                               num_layers=self.num_layers, batch_first=True)
        self.mlp = nn.Sequential(nn.Linear(self.dim_hidden, self.dim_hidden), nn.LayerNorm(self.dim_hidden),
                                 nn.ReLU(), nn.Linear(self.dim_hidden, self.dim_hidden))
        self.norm = nn.LayerNorm(self.dim_hidden)

    _rnn_name = "sup_rnn"

    def post_spec(self):
        return (self.norm, self.mlp[0], self.mlp[1], self.mlp[3])


class Generator(_RecurrentBlock):
    """Noise z -> latent (TimeGANv1.py:254-311): mlp(F->2H, LN, ReLU, 2H->H) -> GRU -> LN -> Linear."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        self.dim_latent = int(cfg['g']['dim_latent'])
        self.dim_hidden = int(cfg['g']['dim_hidden'])
        self.num_layers = int(cfg['g']['num_layers'])
        self.seq_len = int(cfg['system']['seq_len'])
        self.padding_value = int(cfg['system']['padding_value'])
        self.g_rnn = rnn_cls(input_size=self.dim_hidden, hidden_size=self.dim_hidden,
                             num_layers=self.num_layers, batch_first=True)
        self.g_linear = nn.Linear(self.dim_hidden, self.dim_hidden)
        self.mlp = nn.Sequential(nn.Linear(self.dim_latent, self.dim_hidden * 2), nn.LayerNorm(self.dim_hidden * 2),
                                 nn.ReLU(), nn.Linear(self.dim_hidden * 2, self.dim_hidden))
        self.norm = nn.LayerNorm(self.dim_hidden)

    _rnn_name = "g_rnn"

    def pre_spec(self):
        return (None, self.mlp[0], self.mlp[1], self.mlp[3])

    def post_spec(self):
        return (self.norm, self.g_linear, None, None)


class Discriminator(_RecurrentBlock):
    """Latent -> per-step logit (TimeGANv1.py:334-388): GRU -> LN -> Linear(H, 1); output (B, T, 1)."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        self.dim_hidden = int(cfg['d']['dim_hidden'])
        self.num_layers = int(cfg['d']['num_layers'])
        self.seq_len = int(cfg['system']['seq_len'])
        self.padding_value = int(cfg['system']['padding_value'])
        self.d_rnn = rnn_cls(input_size=self.dim_hidden, hidden_size=self.dim_hidden,
                             num_layers=self.num_layers, batch_first=True)
        self.norm = nn.LayerNorm(self.dim_hidden)
        self.mlp = nn.Sequential(nn.Linear(self.dim_hidden, 1))

    _rnn_name = "d_rnn"

    def post_spec(self):
        return (self.norm, self.mlp[0], None, None)


def forward_pair(net_a: _RecurrentBlock, x_a: torch.Tensor, t_a, net_b: _RecurrentBlock, x_b: torch.Tensor, t_b):
    """(net_a(x_a, t_a), net_b(x_b, t_b)) for two networks that do not depend on each other (the embedder
    and the generator of one step, TimeGANv1.py:517-518; recovery and supervisor on the same latent
    sequence, :436-437): their GRU stacks run in ONE kernel launch, forward and backward, side by side
    on the GPU.  Anything the paired kernels do not cover falls back to the two separate calls."""
    rnn_a, rnn_b = getattr(net_a, net_a._rnn_name), getattr(net_b, net_b._rnn_name)
    pairable = (isinstance(rnn_a, GRU) and isinstance(rnn_b, GRU) and x_a.is_cuda and x_b.is_cuda
                and net_a.padding_value == 0 and net_b.padding_value == 0
                and rnn_a.num_layers == rnn_b.num_layers)
    if not pairable:
        return net_a(x_a, t_a), net_b(x_b, t_b)
    pre_a, pre_b = net_a.pre_spec(), net_b.pre_spec()
    if pre_a is not None and pre_b is not None:
        u_a, u_b = head_apply_pair(x_a, pre_a, x_b, pre_b)
    else:
        u_a, u_b = net_a.pre(x_a), net_b.pre(x_b)
    h_a, h_b = gru_pair(rnn_a, u_a, t_a, rnn_b, u_b, t_b)
    return head_apply_pair(h_a, net_a.post_spec(), h_b, net_b.post_spec())


def build_networks(cfg: Dict, rnn_cls=GRU) -> Dict[str, nn.Module]:
    """The five networks in the reference's construction order (TimeGANv1.py:926-930)."""
    return {"emb": Embedding(cfg, rnn_cls), "rec": Recovery(cfg, rnn_cls), "sup": Supervisor(cfg, rnn_cls),
            "g": Generator(cfg, rnn_cls), "d": Discriminator(cfg, rnn_cls)}


def make_cfg(features: int, hidden: int, layers: int, seq_len: int = 24, batch_size: int = 128,
             lr: float = 1e-3, threshold: float = 0.15, device: str = "cuda:0") -> Dict:
    """A cfg dict with the reference's schema (config/config.yaml:1-38)."""
    return {
        "system": {"seq_len": seq_len, "padding_value": 0.0, "lr": lr, "batch_size": batch_size,
                   "device": device, "jointly_num_epochs": 1, "dataset": "synthetic"},
        "emb": {"dim_features": features, "dim_hidden": hidden, "num_layers": layers, "num_epochs": 1},
        "rec": {"dim_output": features, "dim_hidden": hidden, "num_layers": layers},
        "sup": {"dim_hidden": hidden, "num_layers": layers, "num_epochs": 1},
        "g": {"dim_latent": features, "dim_hidden": hidden, "num_layers": layers},
        "d": {"dim_hidden": hidden, "num_layers": layers, "threshold": threshold},
    }
