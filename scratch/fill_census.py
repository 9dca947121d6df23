"""Which Python lines cause aten::fill_/zero_/add/cat launches in one eager joint iteration."""
import sys, collections, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
torch.manual_seed(42)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=False)
x = torch.rand(128, 24, 6, device=dev)
for _ in range(2): tr.joint_step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.joint_step(x)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::cat", "aten::copy_"):
        st = [s for s in ev.stack if "timegan_amd" in s or "time-gan_amd" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, where), n in cnt.most_common(40):
    print(n, name, where[-110:])
