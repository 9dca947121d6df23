"""Joint iterations per second at the stock config (6 features, H 24, 3 layers, T 24, batch 128), graph mode."""
import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, batch_size=128, device='cuda:0')
x = torch.rand(128, 24, 6, device=dev)
torch.manual_seed(42)
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=True)
for _ in range(3): tr.joint_step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): tr.joint_step(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('graphs joint s/iter %.5f  it/s %.2f' % (dt, 1 / dt), flush=True)
