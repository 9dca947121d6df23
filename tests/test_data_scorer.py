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


def test_resident_dataset_all_ranks_take_the_same_number_of_batches():
    """A tail batch with fewer rows than ranks is dropped by every rank (a rank without rows would skip a
    step whose collectives the others wait in)."""
    import timegan_amd as tg
    ds = tg.data.ResidentDataset(np.zeros((18, 2, 3), dtype=np.float32), "cpu")     # batches of 8, 8, 2
    counts = [sum(1 for _ in ds.batches(8, shuffle=False, rank=r, world=4)) for r in range(4)]
    assert counts == [2, 2, 2, 2]
    sizes = [[b.shape[0] for b in ds.batches(5, shuffle=False, rank=r, world=4)] for r in range(4)]   # 5, 5, 5, 3
    assert sizes == [[2, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1]]


def test_fit_runs_the_three_phases_and_model_files_round_trip(tmp_path):
    """fit() = the reference's driver (TimeGANv1.py:940-1009): phases in order, each for its configured
    epochs; save_models / load_models use the reference's file names and plain tensor files."""
    import timegan_amd as tg
    torch.manual_seed(1)
    cfg = tg.make_cfg(features=3, hidden=6, layers=1, seq_len=5, batch_size=4, device="cpu")
    cfg["emb"]["num_epochs"] = 2
    cfg["system"]["dataset"] = "sine"
    tr = tg.TimeGANTrainer(cfg, device="cpu", rnn_cls=torch.nn.GRU)
    ds = tg.data.ResidentDataset(np.random.default_rng(0).uniform(0, 1, (10, 5, 3)).astype(np.float32), "cpu")
    seen = []
    hist = tr.fit(ds, on_batch=lambda phase, epoch, i, out: seen.append((phase, epoch, i)))
    assert [len(hist[k]) for k in ("embedding", "supervisor", "joint")] == [6, 3, 3]      # 3 batches per epoch
    assert seen[0] == ("embedding", 0, 0) and seen[5] == ("embedding", 1, 2) and seen[-1] == ("joint", 0, 2)
    assert set(hist["joint"][0]) >= {"g_loss", "e_loss", "d_loss", "d_stepped"}
    with pytest.raises(ValueError):
        tr.fit(ds, phases=("embedding", "adversarial"))
    paths = tr.save_models(str(tmp_path))
    assert sorted(os.path.basename(p) for p in paths) == ["d_sine.pt", "emb_sine.pt", "g_sine.pt", "rec_sine.pt", "sup_sine.pt"]
    ref = torch.nn.GRU(6, 6, 1, batch_first=True)        # torch's own module takes the stored GRU tensors
    sd = torch.load(os.path.join(str(tmp_path), "sup_sine.pt"), weights_only=True)
    ref.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if "weight_ih_l0" in k or "weight_hh_l0" in k
                         or "bias_ih_l0" in k or "bias_hh_l0" in k})
    tr2 = tg.TimeGANTrainer(cfg, device="cpu", rnn_cls=torch.nn.GRU)
    tr2.load_models(str(tmp_path))
    for k in tr.nets:
        for (n1, p1), (_, p2) in zip(tr.nets[k].state_dict().items(), tr2.nets[k].state_dict().items()):
            assert torch.equal(p1, p2), (k, n1)


@pytest.mark.gpu
def test_fit_with_graphs_equals_eager_fit_including_a_ragged_last_batch():
    """Graph mode replays full batches and runs a batch of another size eagerly; both trainers see the same
    batches and the same noise, so the networks must end up the same."""
    import timegan_amd as tg
    data = np.random.default_rng(0).uniform(0, 1, (44, 8, 5)).astype(np.float32)      # batches of 16, 16, 12
    cfg = tg.make_cfg(features=5, hidden=12, layers=2, seq_len=8, batch_size=16, device="cuda:0")
    cfg["emb"]["num_epochs"] = 2
    cfg["sup"]["num_epochs"] = 2
    cfg["system"]["jointly_num_epochs"] = 2
    results = []
    for graphs in (False, True):
        torch.manual_seed(7)
        tr = tg.TimeGANTrainer(cfg, device="cuda:0", use_graphs=graphs)
        ds = tg.data.ResidentDataset(data, "cuda:0")
        torch.manual_seed(8)           # the noise of the joint phase comes from the global generator
        hist = tr.fit(ds, seed=3)
        results.append((hist, tr.state_dicts()))
    (h0, sd0), (h1, sd1) = results
    assert len(h0["joint"]) == len(h1["joint"]) == 6
    np.testing.assert_allclose(h1["embedding"], h0["embedding"], rtol=2e-4)
    np.testing.assert_allclose(h1["supervisor"], h0["supervisor"], rtol=2e-4)
    for h in (h0, h1):      # the joint phase draws its noise differently in the two modes: sanity only
        assert all(np.isfinite(o["d_loss"]) and np.all(np.isfinite(o["g_loss"])) for o in h["joint"])


def test_water_windows_keep_raw_values_and_drop_three_columns(tmp_path):
    """data/Water.py:23-34: drop columns 0, 1 and the last, no reversal, no scaling, windows [i, i + seq_len) for
    i < len - seq_len.  (The dataset's CSV is not part of the reference checkout; a synthetic table of the same
    layout -- index, time stamp, six sensor columns, event flag -- stands in.)"""
    import numpy as np
    from timegan_amd import data
    rng = np.random.default_rng(0)
    sensors = rng.normal(5.0, 3.0, (40, 6))
    path = tmp_path / "water.csv"
    with open(path, "w") as f:
        f.write("idx,Time,Tp,Cl,pH,Redox,Leit,Trueb,EVENT\n")
        for i, row in enumerate(sensors):
            f.write(f"{i},2017-07-01T00:{i:02d}:00Z," + ",".join(repr(float(v)) for v in row) + ",False\n")
    w = data.load_water_csv(str(path), 24)
    assert w.shape == (16, 24, 6)
    for i in range(16):      # (pandas' default float parser, which the reference uses too, may be an ulp off)
        assert np.allclose(w[i].astype(np.float64), sensors[i:i + 24], rtol=1e-14, atol=0)
    table = np.concatenate([np.arange(40)[:, None], np.zeros((40, 1)), sensors, np.ones((40, 1))], axis=1)
    ww = data.water_windows(table, 24)
    assert ww.shape == w.shape and all(np.array_equal(ww[i], sensors[i:i + 24]) for i in range(16))
    assert data.water_windows(table[:10], 24).shape == (0, 24, 6)
