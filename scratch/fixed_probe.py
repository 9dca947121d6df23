import sys, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from bench import make_weights
dev = torch.device('cuda:0')
T = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, L = 24, 3
w = make_weights(H, H, L, dev, 42)
x = torch.rand(T, B, H, device=dev); dy = torch.randn(T, B, H, device=dev)
r = tg.GRURunner(B, T, H, H, L, tg.LAYOUT_TBF, dev); r.set_weights(w)
for _ in range(100): r.forward(x); r.backward(dy)
torch.cuda.synchronize()
