"""Per-phase time of the mid-H recurrent kernels (diagnostic build libtgru_stamps.so: make -C time-gan_amd/csrc
stamps).  Prints shader-clock-independent microseconds per step for wave 0 of workgroup 0 (C3 layer: B 1024, T 24, H 96)."""
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
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lb = _lib.lib()
lb.tgru_debug_stamps_mid.argtypes = [ctypes.c_void_p]
print('rc', lb.tgru_debug_stamps_mid(out))
a = np.array(out[:], dtype=np.float64) / 100.0 / T       # 100 MHz ticks -> us per step
fn = ['0 state reads + products', '1 drain + result stores', '2 barrier A', '3 gate phase', '4 barrier B', '5', '6', '7 loop top']
bn = ['0 elementwise phase', '1 barrier A', '2 products + stores', '3 barrier B', '4', '5', '6', '7 loop top']
for k in range(8):
    print(f'fwd {fn[k]:30s} {a[k]:7.3f} us')
print('fwd sum', a[:8].sum())
for k in range(8):
    print(f'bwd {bn[k]:30s} {a[8 + k]:7.3f} us')
print('bwd sum', a[8:].sum())
