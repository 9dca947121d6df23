"""How fast is the vendor fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) at the C4 / C3 projection shapes?"""
import torch, time
dev = torch.device('cuda:0')
def t(M, K, N, tb=False, ta=False):
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    f = lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 5; e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"M={M} K={K} N={N} ta={ta} tb={tb}: {ms:.3f} ms  {2*M*K*N/ms/1e9:.1f} TFLOP/s", flush=True)
t(524288, 512, 1536, tb=True)      # gi = X W_ih^T
t(524288, 1536, 512)               # dX = dG W_ih
t(1536, 524288, 512, ta=False)     # dW = dG^T X  (as (1536 x M) . (M x 512)), operand pre-transposed
t(524288, 1536, 512, ta=False)
a = torch.randn(524288, 1536, device=dev); x = torch.randn(524288, 512, device=dev)
for _ in range(2): (a.t() @ x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): (a.t() @ x)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"dW = dG^T X via a.t() @ x: {ms:.3f} ms {2*524288*1536*512/ms/1e9:.1f} TFLOP/s")
t(24576, 96, 288, tb=True)
