"""Host cost of one differentiable GRU call in eager mode at the stock dimensions (B 128, T 24, H 24, L 3):
forward + backward through tg.GRU, against the same work through the preallocated runner (no autograd)."""
import sys, time, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
B, T, H, L = 128, 24, 24, 3
gru = tg.GRU(H, H, L, batch_first=True).to(dev)
x = torch.rand(B, T, H, device=dev, requires_grad=True)
gy = torch.randn(B, T, H, device=dev)
def step():
    gru.zero_grad(set_to_none=True)      # (as a training loop does: the gradients are adopted, not added)
    y, _ = gru(x)
    y.backward(gy)
for _ in range(20): step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"module forward+backward: {(t1 - t0) / n * 1e6:.1f} us per call pair")
with torch.no_grad():
    for _ in range(20): gru(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): gru(x)
    torch.cuda.synchronize()
    print(f"module forward (no_grad): {(time.perf_counter() - t0) / n * 1e6:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
