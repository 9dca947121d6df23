"""Per-phase time of the persistent per-step forward kernel (diagnostic build libtgru_stamps.so:
make -C time-gan_amd/csrc stamps).  Prints microseconds per step for wave 0 of workgroup 0."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from timegan_amd import _lib
import os
_lib.LIB_PATH = _lib.LIB_PATH.replace('libtgru.so', os.environ.get('STAMPS_LIB', 'libtgru_stamps.so'))
from bench import make_weights
dev = torch.device('cuda:0')
B, T, H, L = 2048, 256, 512, 1
r = tg.GRURunner(B, T, H, H, L, tg.LAYOUT_TBF, dev); r.set_weights(make_weights(H, H, L, dev, 1))
x = torch.rand(T, B, H, device=dev); dy = torch.randn(T, B, H, device=dev)
for _ in range(3):
    r.forward(x); r.backward(dy)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lb = _lib.lib()
lb.tgru_debug_stamps_step.argtypes = [ctypes.c_void_p]
print('rc', lb.tgru_debug_stamps_step(out))
a = np.array(out[:], dtype=np.float64) / 100.0 / T       # 100 MHz ticks -> us per step
names = ['0 epilogue operand loads issued', '1 wait for peers', '2 first tiles (ring prologue)', '3 products',
         '4 gates + state tile stores', '5 arrive (vmcnt(0) + barrier + add)', '6 gate / y stores issued', '7 loop top']
for k in range(8):
    print(f'fwd {names[k]:45s} {a[k]:7.2f} us')
print('fwd sum', a[:8].sum())
for k in range(8):
    print(f'bwd phase {k}: {a[8 + k]:7.2f} us')
print('bwd sum', a[8:].sum())
print('(loader-wave kernels: fwd 0 operand loads issued, 1 wait for peers + first tile, 3 products, 4 gates + stores issued, 5 arrival, 6 late stores;\n bwd 0 loads issued, 1 wait for peers + first tile, 2 products, 3 wait for the saved gates, 4 gates read + gradients into LDS, 5 stores + arrival)')
