"""Times the library's GEMM at the stress shapes through diagnostic builds (scratch/gemm_knockout.sh):
gi = X W^T (tgru_input_projection) and dW = dG^T X (tgru_linear_weight_grad)."""
import ctypes, glob, os, sys, torch
dev = torch.device('cuda:0')
here = os.path.dirname(os.path.abspath(__file__))
libs = sorted(glob.glob(os.path.join(here, 'ko', 'libtgru_*.so')))
M, K, N = 524288, 512, 1536
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
dw = torch.empty(N, K, device=dev)
vp = ctypes.c_void_p
def timed(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for path in libs:
    lib = ctypes.CDLL(path)
    lib.tgru_input_projection.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    lib.tgru_linear_weight_grad_workspace_bytes.restype = ctypes.c_size_t
    lib.tgru_linear_weight_grad_workspace_bytes.argtypes = [ctypes.c_int] * 3
    lib.tgru_linear_weight_grad.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    st = vp(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(lib.tgru_linear_weight_grad_workspace_bytes(M, N, K) // 4 + 4, device=dev)
    def gi():
        rc = lib.tgru_input_projection(vp(x.data_ptr()), vp(w.data_ptr()), vp(b.data_ptr()), vp(out.data_ptr()), M, N, K, st)
        assert rc == 0, rc
    def wg():
        rc = lib.tgru_linear_weight_grad(vp(out.data_ptr()), vp(x.data_ptr()), vp(dw.data_ptr()), vp(ws.data_ptr()), M, N, K, 0, st)
        assert rc == 0, rc
    t1 = timed(gi); t2 = timed(wg)
    fl = 2.0 * M * K * N
    print(f"{os.path.basename(path):24s} gi {t1:7.3f} ms {fl/t1/1e9:6.1f} TF   dW {t2:7.3f} ms {fl/t2/1e9:6.1f} TF", flush=True)
