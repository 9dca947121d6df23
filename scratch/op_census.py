"""Which aten ops (and how many) one eager joint iteration issues besides the tgru kernels."""
import sys, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
torch.manual_seed(42)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=("graph" in sys.argv))
x = torch.rand(128, 24, 6, device=dev)
for _ in range(2): tr.joint_step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.joint_step(x)
torch.cuda.synchronize()
rows = [(e.key, e.count) for e in prof.key_averages() if e.key.startswith('aten::') or 'Backward' in e.key or 'Function' in e.key]
rows.sort(key=lambda r: -r[1])
for k, c in rows[:60]:
    print(f"{c:6d}  {k}")
