"""Post-hoc scoring on the HIP GRU: host mirrors of metrics/GeneralRNN.py and of the data flow of
metrics/discriminative.py, plus the two scores that script prepares its arrays for.

The reference's GeneralRNN hard-codes nn.LSTM (GeneralRNN.py:41-44) although its config selects a
cell type (GeneralRNN.py:9-15,38); its scoring script stops after printing the shapes of the split
(discriminative.py:80-87).  Here the recurrent cell is the GRU of this package (type 'gru'), and the
scores follow the TimeGAN protocol the split was written for: a classifier real-vs-synthetic
(discriminative score = |accuracy - 0.5|) and a next-step predictor trained on synthetic data and
tested on real data (predictive score = mean absolute error).  Nothing in the reference pins these
numbers: parity unpinned; the split itself is pinned by tests/golden/scorer_split.npz.
"""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .gru import GRU
from .parallel import all_reduce_grads, broadcast_parameters, dist_sum, multi_rank, shard_bounds


class GeneralRNN(nn.Module):
    """net = RNN (time-major input) -> output sequence -> Linear -> Sigmoid (GeneralRNN.py:27-55)."""

    def __init__(self, cfg: Dict, rnn_cls=GRU):
        super().__init__()
        c = cfg["rnn"]
        self.type = c["type"]
        if self.type != "gru":
            raise NotImplementedError("only the GRU cell runs on the HIP kernels (cfg['rnn']['type'] = 'gru')")
        if str(c.get("bidirectional", "false")).lower() == "true":
            raise NotImplementedError("bidirectional stacks are not used by the reference's call sites")
        self.dim_input = int(c["dim_input"])
        self.dim_output = int(c["dim_output"])
        self.dim_hidden = int(c["dim_hidden"])
        self.num_layers = int(c["num_layers"])
        # dropout between layers is a no-op for one layer (GeneralRNN.py: num_layers 1, dropout 0.3)
        if self.num_layers > 1 and float(c.get("dropout", 0.0)) > 0:
            raise NotImplementedError("inter-layer dropout is not implemented")
        self.rnn = rnn_cls(input_size=self.dim_input, hidden_size=self.dim_hidden, num_layers=self.num_layers)
        self.linear = nn.Linear(self.dim_hidden, self.dim_output)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(self.linear(self.rnn(x)[0]))


def split_real_synthetic(real: np.ndarray, synth: np.ndarray, seq_len: int) -> Dict[str, np.ndarray]:
    """The 75/25 split of discriminative.py:57-78: X = the first seq_len steps, y = the last step."""
    rd = np.asarray(real)
    sd = np.asarray(synth)[:len(rd)]
    n = len(rd)
    idx = np.arange(n)
    n_train = int(.75 * n)
    tr, te = idx[:n_train], idx[n_train:]
    return {"X_stock_train": rd[tr, :seq_len, :], "X_synth_train": sd[tr, :seq_len, :],
            "X_stock_test": rd[te, :seq_len, :], "y_stock_test": rd[te, -1, :],
            "y_stock_train": rd[tr, -1, :], "y_synth_train": sd[tr, -1, :]}


def _fit(model: nn.Module, x: torch.Tensor, y: torch.Tensor, loss_sum, steps: int, batch: int, lr: float,
         generator: torch.Generator, group=None) -> None:
    """Adam on minibatches drawn with `generator` (every rank holds the same generator state, hence the
    same minibatch).  `loss_sum(out, target)` returns the SUM of the per-element losses and their count;
    with several ranks each takes a contiguous shard of the minibatch, the loss is that sum over the
    GLOBAL count and the gradients are summed over ranks: the update equals the single-process one."""
    broadcast_parameters([model], group=group)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    n = x.shape[0]
    for _ in range(steps):
        idx = torch.randint(0, n, (min(batch, n),), generator=generator)
        lo, hi = shard_bounds(idx.numel(), group)
        idx = idx[lo:hi].to(x.device)
        opt.zero_grad()
        if idx.numel():
            total, count = loss_sum(model(x[idx].transpose(0, 1).contiguous()), y[idx])
        else:       # more ranks than rows in this minibatch: contribute nothing, keep the collectives aligned
            total, count = sum(p.sum() for p in model.parameters()) * 0.0, 0
        count = dist_sum(torch.tensor(float(count), device=x.device), group)
        (total / count).backward()
        all_reduce_grads([model], group=group)
        opt.step()


def discriminative_score(real: np.ndarray, synth: np.ndarray, device, hidden: int = 24, steps: int = 200,
                         batch: int = 128, lr: float = 1e-3, seed: int = 0, group=None, rnn_cls=GRU) -> float:
    """|accuracy - 0.5| of a GRU classifier real (1) vs synthetic (0), trained on 75 % of the pairs
    and evaluated on the rest.  0 = indistinguishable.  With a process group every rank passes the same
    arrays and gets the same score: minibatches and the test set are sharded over the ranks."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    real_t = torch.as_tensor(np.asarray(real), dtype=torch.float32, device=device)
    synth_t = torch.as_tensor(np.asarray(synth)[:len(real)], dtype=torch.float32, device=device)
    n = real_t.shape[0]
    n_train = int(.75 * n)
    cfg = {"rnn": {"type": "gru", "dim_input": real_t.shape[2], "dim_output": 1, "dim_hidden": hidden,
                   "num_layers": 1, "dropout": 0.0, "bidirectional": "false"}}
    clf = GeneralRNN(cfg, rnn_cls=rnn_cls).to(device)
    x = torch.cat([real_t[:n_train], synth_t[:n_train]])
    y = torch.cat([torch.ones(n_train, device=device), torch.zeros(n_train, device=device)])
    bce = nn.BCELoss(reduction="sum")
    _fit(clf, x, y, lambda out, tgt: (bce(out[-1, :, 0], tgt), tgt.numel()), steps, batch, lr, g, group)
    with torch.no_grad():
        xt = torch.cat([real_t[n_train:], synth_t[n_train:]])
        yt = torch.cat([torch.ones(n - n_train), torch.zeros(n - n_train)]).to(device)
        lo, hi = shard_bounds(xt.shape[0], group)
        hits = xt.new_zeros(())
        if hi > lo:
            hits = ((clf(xt[lo:hi].transpose(0, 1).contiguous())[-1, :, 0] > 0.5).float() == yt[lo:hi]).float().sum()
        acc = (dist_sum(hits, group) / xt.shape[0]).item()
    return abs(acc - 0.5)


def predictive_score(real: np.ndarray, synth: np.ndarray, device, hidden: int = 24, steps: int = 200,
                     batch: int = 128, lr: float = 1e-3, seed: int = 0, group=None, rnn_cls=GRU) -> float:
    """Mean absolute error, on the real test windows, of a GRU trained on synthetic windows to predict
    the last step of a window (y of discriminative.py:74-78) from the steps before it.  Sharded over
    the ranks of `group` like discriminative_score."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    sp = split_real_synthetic(real, synth, np.asarray(real).shape[1])
    f = sp["X_synth_train"].shape[2]
    cfg = {"rnn": {"type": "gru", "dim_input": f, "dim_output": f, "dim_hidden": hidden,
                   "num_layers": 1, "dropout": 0.0, "bidirectional": "false"}}
    model = GeneralRNN(cfg, rnn_cls=rnn_cls).to(device)
    xs = torch.as_tensor(sp["X_synth_train"][:, :-1, :], dtype=torch.float32, device=device)
    ys = torch.as_tensor(sp["y_synth_train"], dtype=torch.float32, device=device)
    l1 = nn.L1Loss(reduction="sum")
    _fit(model, xs, ys, lambda out, tgt: (l1(out[-1], tgt), tgt.numel()), steps, batch, lr, g, group)
    with torch.no_grad():
        xr = torch.as_tensor(sp["X_stock_test"][:, :-1, :], dtype=torch.float32, device=device)
        yr = torch.as_tensor(sp["y_stock_test"], dtype=torch.float32, device=device)
        lo, hi = shard_bounds(xr.shape[0], group)
        err = xr.new_zeros(())
        if hi > lo:
            err = l1(model(xr[lo:hi].transpose(0, 1).contiguous())[-1], yr[lo:hi])
        return (dist_sum(err, group) / yr.numel()).item()
