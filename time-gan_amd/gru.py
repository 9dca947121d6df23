"""`GRU`: a drop-in for torch.nn.GRU as the reference uses it, running on libtgru.so.

Constructor arguments, parameter names / shapes / gate order (r, z, n), default initialisation
and the forward signature follow torch.nn.GRU (torch/nn/modules/rnn.py:1182-1454) as exercised at
outdated_files/TimeGANv1.py:35-38,61 and metrics/GeneralRNN.py:41-44, so state_dicts written at
outdated_files/TimeGANv1.py:953-954,976,1006,1009 load into it and vice versa.
"""
import math
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence

from . import functional as F_
from ._lib import LAYOUT_BTF, LAYOUT_PACKED, LAYOUT_TBF


class GRU(nn.Module):
    # Eager calls go through the C++ autograd binding when it is built (functional.cpp_op); the recurrent blocks of
    # timegan_amd.modules switch it off for their GRUs, whose gradients go through the deferred-reduction queue.
    cpp_op = True

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 batch_first: bool = False, dropout: float = 0.0, bidirectional: bool = False,
                 device=None, dtype=None):
        super().__init__()
        if bidirectional:
            raise NotImplementedError("bidirectional GRUs are not used by the reference and not built")
        if not bias:
            raise NotImplementedError("bias=False is not used by the reference and not built")
        if dtype not in (None, torch.float32):
            raise NotImplementedError("the GRU kernels compute in float32")
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bias = bias
        self.batch_first = batch_first
        self.dropout = float(dropout)
        self.bidirectional = False
        self._flat_names = []
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else hidden_size
            shapes = {"weight_ih": (3 * hidden_size, in_l), "weight_hh": (3 * hidden_size, hidden_size),
                      "bias_ih": (3 * hidden_size,), "bias_hh": (3 * hidden_size,)}
            for name, shape in shapes.items():
                pname = f"{name}_l{layer}"
                setattr(self, pname, nn.Parameter(torch.empty(shape, device=device, dtype=torch.float32)))
                self._flat_names.append(pname)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        # same distribution and the same draw order as torch (rnn.py:296-299)
        stdv = 1.0 / math.sqrt(self.hidden_size) if self.hidden_size > 0 else 0
        for weight in self.parameters():
            nn.init.uniform_(weight, -stdv, stdv)

    def extra_repr(self) -> str:
        return (f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, "
                f"batch_first={self.batch_first}")

    @property
    def _flat_weights(self):
        return [getattr(self, n) for n in self._flat_names]

    def _check_dropout(self):
        if self.dropout > 0 and self.num_layers > 1 and self.training:
            raise NotImplementedError("inter-layer dropout is not used by the reference and not built")

    def forward(self, input: Union[torch.Tensor, PackedSequence], hx: Optional[torch.Tensor] = None,
                lengths=None) -> Tuple[Union[torch.Tensor, PackedSequence], torch.Tensor]:
        """nn.GRU.forward, plus an optional `lengths` for padded tensors.

        With a padded tensor and `lengths` (non-increasing, as enforce_sorted=True demands) the result
        equals pack_padded_sequence -> GRU -> pad_packed_sequence(padding_value=0,
        total_length=T): rows past a sequence's length are zero and receive zero gradient.
        """
        self._check_dropout()
        weights = self._flat_weights
        L, H = self.num_layers, self.hidden_size
        if isinstance(input, PackedSequence):
            data, batch_sizes, sorted_idx, unsorted_idx = input
            T = int(batch_sizes.numel())
            B = int(batch_sizes[0])
            if hx is not None and sorted_idx is not None:
                hx = hx.index_select(1, sorted_idx)
            if int(batch_sizes[-1]) == B:
                # every sequence is full length: packed rows are exactly (T,B,In) time-major
                y, h_n = F_.gru_apply(data.view(T, B, -1), weights, B, T, LAYOUT_TBF, h0=hx, cpp=self.cpp_op)
                y = y.view(T * B, H)
            else:
                bs = batch_sizes.to(torch.int64)
                lens = (bs.unsqueeze(0) > torch.arange(B).unsqueeze(1)).sum(dim=1).to(torch.int32)
                toff = torch.zeros(T + 1, dtype=torch.int32)
                toff[1:] = torch.cumsum(bs, 0).to(torch.int32)
                y, h_n = F_.gru_apply(data, weights, B, T, LAYOUT_PACKED, h0=hx,
                                      lengths=lens.to(data.device), toff=toff.to(data.device), cpp=self.cpp_op)
            if unsorted_idx is not None:
                h_n = h_n.index_select(1, unsorted_idx)
            return PackedSequence(y, batch_sizes, sorted_idx, unsorted_idx), h_n

        if input.dim() == 2:   # unbatched (T, In)
            x = input.unsqueeze(1 if not self.batch_first else 0)
            hx2 = None if hx is None else hx.unsqueeze(1)
            y, h_n = self.forward(x, hx2, lengths)
            return y.squeeze(1 if not self.batch_first else 0), h_n.squeeze(1)
        if input.dim() != 3:
            raise ValueError(f"GRU: Expected input to be 2D or 3D, got {input.dim()}D instead")
        if input.shape[-1] != self.input_size:
            raise RuntimeError(f"input.size(-1) must be equal to input_size. Expected {self.input_size}, "
                               f"got {input.shape[-1]}")
        if self.batch_first:
            B, T = input.shape[0], input.shape[1]
            layout = LAYOUT_BTF
        else:
            T, B = input.shape[0], input.shape[1]
            layout = LAYOUT_TBF
        if hx is not None and tuple(hx.shape) != (L, B, H):
            raise RuntimeError(f"Expected hidden size {(L, B, H)}, got {list(hx.shape)}")
        lens_dev = F_.lengths_to_device(lengths, B, T, input.device)
        return F_.gru_apply(input, weights, B, T, layout, h0=hx, lengths=lens_dev, cpp=self.cpp_op)


def gru_pair(rnn_a: "GRU", x_a: torch.Tensor, lengths_a, rnn_b: "GRU", x_b: torch.Tensor, lengths_b):
    """(y_a, y_b) = (rnn_a(x_a, lengths=lengths_a)[0], rnn_b(x_b, lengths=lengths_b)[0]) for two INDEPENDENT
    stacks, evaluated together: with the same number of layers and hidden sizes of at most 32 both run in
    one kernel launch, forward and backward (F_.gru_apply_pair).  Padded (B,T,F) / (T,B,F) inputs only."""
    calls = []
    for rnn, x, lengths in ((rnn_a, x_a, lengths_a), (rnn_b, x_b, lengths_b)):
        rnn._check_dropout()
        if x.dim() != 3 or x.shape[-1] != rnn.input_size:
            raise RuntimeError(f"gru_pair: expected a 3D input with {rnn.input_size} features, got {tuple(x.shape)}")
        if rnn.batch_first:
            B, T, layout = x.shape[0], x.shape[1], LAYOUT_BTF
        else:
            T, B, layout = x.shape[0], x.shape[1], LAYOUT_TBF
        calls.append((x, rnn._flat_weights, B, T, layout, None, F_.lengths_to_device(lengths, B, T, x.device), None))
    if rnn_a.num_layers != rnn_b.num_layers:
        return rnn_a(x_a, lengths=lengths_a)[0], rnn_b(x_b, lengths=lengths_b)[0]
    (y_a, _), (y_b, _) = F_.gru_apply_pair(calls[0], calls[1])
    return y_a, y_b
