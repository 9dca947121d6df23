"""The LDS-resident head kernels at the energy config's shapes: a few forward + backward calls each (for counter passes)."""
import sys, torch, torch.nn as nn
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
torch.manual_seed(0)
def run(M, din, dmid, dout, deep, ln0, reps=4):
    x = torch.randn(M, din, device=dev, requires_grad=True)
    l0 = nn.LayerNorm(din).to(dev) if ln0 else None
    lin1 = nn.Linear(din, dmid).to(dev)
    l1 = nn.LayerNorm(dmid).to(dev) if deep else None
    lin2 = nn.Linear(dmid, dout).to(dev) if deep else None
    for _ in range(reps):
        y = tg.head_apply(x, l0, lin1, l1, lin2)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
run(24576, 96, 96, 96, True, True)
run(24576, 28, 192, 96, True, False)
print("done")
