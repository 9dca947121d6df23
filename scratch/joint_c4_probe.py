"""One embedding step, one supervisor step and one joint iteration at the stress dimensions (64 features, H 512,
3 layers, T 256) on one GPU, after one warm-up of each; batch from the command line (default 2048, the per-GPU
shard of the stress configuration)."""
import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda:0')
cfg = tg.make_cfg(features=64, hidden=512, layers=3, seq_len=256, batch_size=B, device='cuda:0')
x = torch.rand(B, 256, 64, device=dev)
torch.manual_seed(42)
tr = tg.TimeGANTrainer(cfg, device=dev)
for name, fn in (("embedding", tr.embedding_step), ("supervisor", tr.supervisor_step), ("joint", tr.joint_step)):
    fn(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{name}: {dt:.4f} s/iter', flush=True)
print('peak memory GiB %.1f' % (torch.cuda.max_memory_allocated(dev) / 2 ** 30), 'timeouts', tg.persist_timeouts(), flush=True)
