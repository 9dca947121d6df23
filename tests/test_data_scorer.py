"""Data generators / window loaders / scorer split against fixtures produced by the reference's own
data classes (tests/golden/scorer_split.npz), and the scorer networks on the HIP GRU."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture():
    return np.load(os.path.join(GOLDEN, "scorer_split.npz"))


def test_sine_wave_matches_reference_draw_order():
    import timegan_amd as tg
    d = _fixture()
    got = tg.data.sine_wave(40, 24, 6, rng=np.random.RandomState(int(d["sine.seed"])))
    assert got.shape == d["sine.data"].shape
    np.testing.assert_allclose(got, d["sine.data"], rtol=0, atol=1e-15)
    assert got.min() >= 0.0 and got.max() <= 1.0


def test_windows_and_minmax_match_reference():
    import timegan_amd as tg
    d = _fixture()
    got = tg.data.sliding_windows(d["table"], 24)
    assert got.shape == d["energy.windows"].shape == (36, 24, 4)
    np.testing.assert_allclose(got, d["energy.windows"], rtol=0, atol=1e-15)
    t, msl = tg.data.extract_time(got[:5])
    assert t == list(d["energy.time"]) and msl == int(d["energy.max_seq_len"])
    assert tg.data.sliding_windows(d["table"][:10], 24).shape == (0, 24, 4)


def test_split_matches_reference_indexing():
    import timegan_amd as tg
    d = _fixture()
    sp = tg.scorer.split_real_synthetic(d["sine.data"], d["split.synth"], 24)
    assert sp["X_stock_train"].shape == (30, 24, 6) and sp["X_stock_test"].shape == (10, 24, 6)
    for k, v in sp.items():
        np.testing.assert_array_equal(v, d["split." + k])


def test_resident_dataset_shards_cover_every_sample_once():
    import timegan_amd as tg
    data = np.arange(23 * 2 * 3, dtype=np.float32).reshape(23, 2, 3)
    ds = tg.data.ResidentDataset(data, "cpu")
    seen = []
    for rank in range(2):
        g = torch.Generator().manual_seed(5)
        for b in ds.batches(8, shuffle=True, generator=g, rank=rank, world=2):
            seen.append(b[:, 0, 0].numpy())
    seen = np.sort(np.concatenate(seen))
    np.testing.assert_array_equal(seen, data[:, 0, 0])


def test_general_rnn_rejects_what_the_kernels_do_not_run():
    import timegan_amd as tg
    cfg = {"rnn": {"type": "lstm", "dim_input": 28, "dim_output": 1, "dim_hidden": 256, "num_layers": 1,
                   "dropout": 0.3, "bidirectional": "false"}}
    with pytest.raises(NotImplementedError):
        tg.scorer.GeneralRNN(cfg)


@pytest.mark.gpu
def test_general_rnn_gru_matches_torch_cpu():
    """The scorer network at the reference's sizes (28 -> 256, one layer, time-major, GeneralRNN.py:
    57-75) with the GRU cell: HIP path against the same network on torch's CPU GRU."""
    import timegan_amd as tg
    from helpers import FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close
    cfg = {"rnn": {"type": "gru", "dim_input": 28, "dim_output": 1, "dim_hidden": 256, "num_layers": 1,
                   "dropout": 0.3, "bidirectional": "false"}}
    torch.manual_seed(3)
    ref = tg.scorer.GeneralRNN(cfg, rnn_cls=torch.nn.GRU)
    ours = tg.scorer.GeneralRNN(cfg).to("cuda:0")
    ours.load_state_dict(ref.state_dict())
    x = torch.randn(24, 5, 28)
    yr = ref(x)
    yo = ours(x.cuda())
    assert tuple(yo.shape) == (24, 5, 1)
    yr.sum().backward()
    yo.sum().backward()
    assert_close(yo.detach().cpu().numpy(), yr.detach().numpy(), FWD_RTOL, FWD_ATOL_SCALE, "scorer y")
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert_close(po.grad.cpu().numpy(), pr.grad.numpy(), GRAD_RTOL, GRAD_ATOL_SCALE, name)


@pytest.mark.gpu
def test_scores_separate_good_from_bad_synthetic_data():
    """Parity unpinned (the reference computes no score): sanity of the protocol only -- synthetic data
    drawn from the real generator must score better than uniform noise on both scores."""
    import timegan_amd as tg
    real = tg.data.sine_wave(400, 24, 5, rng=np.random.RandomState(0))
    good = tg.data.sine_wave(400, 24, 5, rng=np.random.RandomState(1))
    bad = np.random.RandomState(2).rand(400, 24, 5)
    dev = torch.device("cuda:0")
    d_good = tg.scorer.discriminative_score(real, good, dev, steps=300)
    d_bad = tg.scorer.discriminative_score(real, bad, dev, steps=300)
    p_good = tg.scorer.predictive_score(real, good, dev, steps=300)
    p_bad = tg.scorer.predictive_score(real, bad, dev, steps=300)
    assert 0.0 <= d_good <= 0.5 and 0.0 <= d_bad <= 0.5
    assert d_bad > 0.4 and d_good < d_bad
    assert p_good < p_bad
