import sys, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from bench import make_weights
dev = torch.device('cuda:0')
def run(B, T, H=24, L=3, n=200):
    w = make_weights(H, H, L, dev, 42)
    x = torch.rand(T, B, H, device=dev); dy = torch.randn(T, B, H, device=dev)
    r = tg.GRURunner(B, T, H, H, L, tg.LAYOUT_TBF, dev); r.set_weights(w)
    for _ in range(20): r.forward(x); r.backward(dy)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        e[0].record(); r.forward(x); e[1].record(); r.backward(dy); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"B={B} T={T} H={H} L={L}: fwd {1e3*tf/n:.1f} us  bwd(+reduce) {1e3*tb/n:.1f} us", flush=True)
for T in (24, 48, 96):
    run(128, T)
for B in (32, 256, 384, 512, 1024, 2048):
    run(B, 24)
