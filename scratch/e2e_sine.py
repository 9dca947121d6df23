"""End-to-end check on one GPU: three-phase training on the sine data (5 features, T 24, H 24, 3 layers,
batch 128) with one HIP graph per step, then generation and the two post-hoc scores, before and after
training.  Prints iterations per second of each phase and the scores."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import timegan_amd as tg

dev = torch.device('cuda:0')
torch.manual_seed(0)
np.random.seed(0)
real = np.asarray(tg.data.sine_wave(samples_number=2048, seq_len=24, features_dim=5), dtype=np.float32)
cfg = tg.make_cfg(features=5, hidden=24, layers=3, seq_len=24, batch_size=128, device='cuda:0')
cfg["emb"]["num_epochs"], cfg["sup"]["num_epochs"], cfg["system"]["jointly_num_epochs"] = 40, 40, 40
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=True)
ds = tg.data.ResidentDataset(real, dev)

def scores(tag):
    synth = tr.generate_all(len(real), seed=1).cpu().numpy()
    d = tg.scorer.discriminative_score(real, synth, dev, steps=400)
    p = tg.scorer.predictive_score(real, synth, dev, steps=400)
    print(f"{tag}: discriminative {d:.3f}  predictive {p:.4f}", flush=True)

scores("untrained")
for phase in ("embedding", "supervisor", "joint"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = tr.fit(ds, phases=(phase,), seed=2)[phase]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    last = h[-1] if not isinstance(h[-1], dict) else {k: (np.mean(v) if isinstance(v, list) else v) for k, v in h[-1].items()}
    print(f"{phase}: {len(h)} iterations in {dt:.2f} s = {len(h)/dt:.0f} it/s, first {h[0] if not isinstance(h[0], dict) else h[0]['d_loss']}, last {last}", flush=True)
scores("trained")
