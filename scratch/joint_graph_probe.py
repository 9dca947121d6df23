import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
torch.manual_seed(42)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=True)
x = torch.rand(128, 24, 6, device=dev)
for _ in range(3): tr.joint_step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 4
for _ in range(n): tr.joint_step(x)
torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/n
print('joint s/iter', dt, 'it/s', 1/dt)
