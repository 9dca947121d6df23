"""Per-phase shader cycles per step of the fused H=96 forward kernel (diagnostic build libtgru_stamps.so):
wave 0 (recurrent product + gates) and wave 5 (input projection) of workgroup 0; C3 layer B 1024, T 24."""
import sys, ctypes, numpy as np, torch, os
sys.path.insert(0, '.')
import timegan_amd as tg
from timegan_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libtgru.so', os.environ.get('STAMPS_LIB', 'libtgru_stamps.so'))
from bench import make_weights
dev = torch.device('cuda:0')
B, T, H, L = 1024, 24, 96, 1
r = tg.GRURunner(B, T, H, H, L, tg.LAYOUT_TBF, dev); r.set_weights(make_weights(H, H, L, dev, 1))
x = torch.rand(T, B, H, device=dev); dy = torch.randn(T, B, H, device=dev)
for _ in range(3):
    r.forward(x); r.backward(dy)
r.forward(x)          # (the backward kernel writes the same stamp array: the forward kernel runs last)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lb = _lib.lib()
lb.tgru_debug_stamps_mid96.argtypes = [ctypes.c_void_p]
print('rc', lb.tgru_debug_stamps_mid96(out))
a = np.array(out[:], dtype=np.float64) / T
names = ['0 loop top', '1 (after product)', '2 barrier A', '3 gates | input rows', '4 barrier B', '5 product: reads + MFMAs', '6 product: reduce + stores', '7']
for k in range(7):
    print(f'{names[k]:24s} wave 0 {a[k]:8.0f}   wave 5 {a[8 + k]:8.0f}  cycles/step')
print('sum', a[:7].sum(), a[8:15].sum())
print('prologue (kernel start -> matrix rows in registers), cycles: wave 0', a[7] * T, ' wave 5', a[15] * T)
