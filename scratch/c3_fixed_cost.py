"""Fixed cost per launch of the H = 96 kernels: the C3 stack (B 1024, H 96, L 3) at T = 12, 24, 48 -- the time that
does not scale with T is prologue + epilogue + launch."""
import sys, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from bench import make_weights
dev = torch.device('cuda:0')
B, H, L = 1024, 96, 3
res = {}
for T in (12, 24, 48, 96):
    r = tg.GRURunner(B, T, H, H, L, tg.LAYOUT_TBF, dev); r.set_weights(make_weights(H, H, L, dev, 1))
    x = torch.rand(T, B, H, device=dev); dy = torch.randn(T, B, H, device=dev)
    for _ in range(10):
        r.forward(x); r.backward(dy)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 100
    tf = tb = 0.0
    for _ in range(n):
        e[0].record(); r.forward(x); e[1].record(); r.backward(dy); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    res[T] = (tf / n * 1e3, tb / n * 1e3)
    print(f"T {T:3d}: forward {res[T][0]:7.1f} us, backward {res[T][1]:7.1f} us")
for a, b in ((12, 24), (24, 48), (48, 96)):
    sf = (res[b][0] - res[a][0]) / (b - a); sb = (res[b][1] - res[a][1]) / (b - a)
    print(f"T {a}->{b}: per step fwd {sf:.2f} us bwd {sb:.2f} us; fixed fwd {res[a][0] - sf * a:.1f} us bwd {res[a][1] - sb * a:.1f} us (3 layers)")
