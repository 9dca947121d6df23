import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
torch.manual_seed(42)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
tr = tg.TimeGANTrainer(cfg, device=dev)
x = torch.rand(128, 24, 6, device=dev)
for name, fn in (('embedding', tr.embedding_step), ('supervisor', tr.supervisor_step)):
    for _ in range(3): fn(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn(x)
    torch.cuda.synchronize(); print(name, 'it/s', 20/(time.perf_counter()-t0))
for _ in range(2): tr.joint_step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): tr.joint_step(x)
torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/5
print('joint s/iter', dt, 'it/s', 1/dt)
