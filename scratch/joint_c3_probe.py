"""Joint iterations per second at the energy-like config (28 features, H 96, 3 layers, T 24, batch 1024)."""
import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
cfg = tg.make_cfg(features=28, hidden=96, layers=3, seq_len=24, batch_size=1024, device='cuda:0')
x = torch.rand(1024, 24, 28, device=dev)
for graphs in (False, True):
    torch.manual_seed(42)
    tr = tg.TimeGANTrainer(cfg, device=dev, use_graphs=graphs)
    for _ in range(3): tr.joint_step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): tr.joint_step(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('graphs', graphs, 'joint s/iter %.4f  it/s %.2f' % (dt, 1 / dt), flush=True)
