"""Host-side profile (cProfile) of one eager joint iteration: where the Python time goes."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
torch.manual_seed(42)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=False)
x = torch.rand(128, 24, 6, device=dev)
for _ in range(3): tr.joint_step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): tr.joint_step(x)
torch.cuda.synchronize(); print('eager it/s', 3 / (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.joint_step(x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
