import os, sys, faulthandler, torch, torch.distributed as dist
sys.path.insert(0, '.')
rank = int(os.environ['RANK'])
log = open(f'gpurun_out/dp_rank{rank}.log', 'w')
faulthandler.dump_traceback_later(45, file=log, exit=True)
import timegan_amd as tg
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group(sys.argv[1] if len(sys.argv) > 1 else 'gloo')
mode = sys.argv[2] if len(sys.argv) > 2 else 'plain'
if mode != 'plain':
    g = torch.zeros(1000, device=dev)
    for _ in range(25): dist.all_reduce(g)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    if mode == 'f64':
        tmax = torch.tensor([1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX); print('tmax', float(tmax.item()), file=log, flush=True)
cfg = tg.make_cfg(features=6, hidden=24, layers=3, seq_len=24, device='cuda:0')
torch.manual_seed(42)
tr = tg.TimeGANTrainer(cfg, device=dev, group=dist.group.WORLD)
x = torch.rand(128, 24, 6, device=dev)
for name, fn in (('emb', tr.embedding_step), ('sup', tr.supervisor_step), ('joint', tr.joint_step)):
    for i in range(2):
        out = fn(x)
        print(name, i, 'ok', file=log, flush=True)
dist.barrier()
print('done', file=log, flush=True)
