"""Data side of the TimeGAN path, host mirrors of data/SineWave.py, data/Energy.py and data/Stock.py.

The reference builds Python lists of per-sample arrays and feeds them through a DataLoader with ten
worker processes; here a dataset is ONE array (numpy on the host, or a tensor resident in HBM) and a
batch is a slice of it, so the input pipeline cannot cap the iteration rate.
"""
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch


def min_max_scale(data: np.ndarray) -> np.ndarray:
    """MinMaxScaler of data/Energy.py:5-9 (per column, 1e-7 in the denominator)."""
    lo = np.min(data, 0)
    return (data - lo) / (np.max(data, 0) - lo + 1e-7)


def sliding_windows(table: np.ndarray, seq_len: int) -> np.ndarray:
    """The window construction of Energy / Stock (data/Energy.py:25-31): reverse the rows, scale,
    then every window [i, i + seq_len) for i < len - seq_len.  Returns (N, seq_len, F)."""
    scaled = min_max_scale(np.asarray(table)[::-1])
    n = len(scaled) - seq_len
    if n <= 0:
        return np.empty((0, seq_len, scaled.shape[1]), dtype=scaled.dtype)
    idx = np.arange(n)[:, None] + np.arange(seq_len)[None, :]
    return scaled[idx]


def load_csv_windows(path: str, seq_len: int) -> np.ndarray:
    """Energy(seq_len, path) / Stock(seq_len, path): a CSV with one header row."""
    return sliding_windows(np.loadtxt(path, delimiter=",", skiprows=1), seq_len)


def water_windows(table: np.ndarray, seq_len: int) -> np.ndarray:
    """The window construction of the water-quality dataset (data/Water.py:23-34) on the table as read from its
    CSV: the first two columns and the last one are dropped, the rows keep their order and their RAW values (no
    reversal and no MinMax scaling, unlike Energy / Stock), then every window [i, i + seq_len) for
    i < len - seq_len.  Returns (N, seq_len, columns - 3)."""
    data = np.asarray(table)[:, 2:-1]
    n = len(data) - seq_len
    if n <= 0:
        return np.empty((0, seq_len, data.shape[1]), dtype=data.dtype)
    idx = np.arange(n)[:, None] + np.arange(seq_len)[None, :]
    return data[idx]


def load_water_csv(path: str, seq_len: int) -> np.ndarray:
    """Water(seq_len, path) (data/Water.py:23-34): a CSV with one header row whose first two columns (an index and
    a time stamp) and last column (the event label) are not features; read with pandas as the reference does, so
    that a non-numeric time stamp column is no obstacle."""
    import pandas as pd
    frame = pd.read_csv(path)
    frame = frame.drop([frame.columns[0], frame.columns[1], frame.columns[-1]], axis=1)
    table = frame.to_numpy()
    n = len(table) - seq_len
    if n <= 0:
        return np.empty((0, seq_len, table.shape[1]), dtype=table.dtype)
    idx = np.arange(n)[:, None] + np.arange(seq_len)[None, :]
    return table[idx]


def sine_wave(samples_number: int, seq_len: int, features_dim: int,
              rng: Optional[np.random.RandomState] = None) -> np.ndarray:
    """SineWave(samples_number, seq_len, features_dim).get_distribution() as one (N, T, F) array
    (data/SineWave.py:5-25): per sample and feature a frequency and a phase from U(0, 0.1) -- drawn
    in the reference's order, so the same legacy seed gives the same data -- then
    (sin(freq * j + phase) + 1) / 2."""
    rng = np.random if rng is None else rng
    fp = rng.uniform(0, 0.1, size=(samples_number, features_dim, 2))
    j = np.arange(seq_len, dtype=np.float64)[None, :, None]
    freq = fp[:, None, :, 0]
    phase = fp[:, None, :, 1]
    return (np.sin(freq * j + phase) + 1) * 0.5


def extract_time(data) -> Tuple[List[int], int]:
    """extract_time of data/Energy.py:12-19 for a dense batch: every length is the time dimension."""
    n, t = int(data.shape[0]), int(data.shape[1])
    return [t] * n, t


class ResidentDataset:
    """A dataset held as one float32 tensor on the device; batches are index gathers in HBM."""

    def __init__(self, samples, device):
        self.data = torch.as_tensor(np.asarray(samples), dtype=torch.float32).to(device)

    def __len__(self) -> int:
        return int(self.data.shape[0])

    def batches(self, batch_size: int, shuffle: bool = True, generator: Optional[torch.Generator] = None,
                rank: int = 0, world: int = 1, with_extent: bool = False) -> Iterator[torch.Tensor]:
        """Batches of one epoch; with `world` > 1 every rank takes its contiguous shard of each global
        batch (the same permutation must be drawn on every rank: pass identically seeded generators).
        `with_extent`: yield (batch, (start, stop, rows of the global batch)) -- where the shard sits in its
        global batch, known here by arithmetic."""
        n = len(self)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        order = order.to(self.data.device)
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            extent = (0, int(idx.numel()), int(idx.numel()))
            if world > 1:
                # shard sizes differ by at most one (parallel.shard_batch); a tail batch with fewer rows than
                # ranks is dropped by EVERY rank, so that all ranks run the same number of steps
                if idx.numel() < world:
                    continue
                base, rem = divmod(idx.numel(), world)
                start = rank * base + min(rank, rem)
                extent = (start, start + base + (1 if rank < rem else 0), int(idx.numel()))
                idx = idx[start:extent[1]]
            if idx.numel():
                x = self.data.index_select(0, idx)
                yield (x, extent) if with_extent else x
