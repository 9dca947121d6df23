"""Times the fused head kernels (forward, backward + reduce) at the joint-training shapes."""
import sys, torch, torch.nn as nn
sys.path.insert(0, '.')
import timegan_amd as tg
import os
if os.environ.get('TGRU_LIB'):
    import timegan_amd._lib as _l
    _l.LIB_PATH = os.environ['TGRU_LIB']
dev = torch.device('cuda:0')
torch.manual_seed(0)
def bench(M, din, dmid, dout, deep, ln0):
    x = torch.randn(M, din, device=dev, requires_grad=True)
    l0 = nn.LayerNorm(din).to(dev) if ln0 else None
    lin1 = nn.Linear(din, dmid).to(dev)
    l1 = nn.LayerNorm(dmid).to(dev) if deep else None
    lin2 = nn.Linear(dmid, dout).to(dev) if deep else None
    y = tg.head_apply(x, l0, lin1, l1, lin2)
    dy = torch.randn_like(y)
    def f(): return tg.head_apply(x, l0, lin1, l1, lin2)
    def b(y): y.backward(dy)
    for _ in range(5): b(f())
    n = 50
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3 * n)]
    for i in range(n):
        e[3*i].record(); y = f(); e[3*i+1].record(); b(y); e[3*i+2].record()
    torch.cuda.synchronize()
    fw = sorted(e[3*i].elapsed_time(e[3*i+1]) for i in range(n))[n//2]
    bw = sorted(e[3*i+1].elapsed_time(e[3*i+2]) for i in range(n))[n//2]
    print(f"M={M} {din}->{dmid}->{dout if deep else '-'} ln0={ln0}: fwd {fw*1e3:.1f} us  bwd {bw*1e3:.1f} us (event-to-event, includes launch gaps)")
bench(3072, 24, 24, 24, True, True)
bench(3072, 24, 24, 6, True, True)
bench(3072, 6, 24, 24, True, False)
bench(3072, 24, 24, 24, False, True)
bench(6144, 24, 24, 24, True, True)
bench(3072, 6, 48, 24, True, False)
