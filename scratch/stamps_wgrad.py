"""Shader cycles of the mid-H weight-gradient kernel (diagnostic build): prologue, stage loop, epilogue of wave 0 of
workgroup 0 at the C3 layer shape."""
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
out = (ctypes.c_ulonglong * 4)()
lb = _lib.lib()
lb.tgru_debug_stamps_wgrad.argtypes = [ctypes.c_void_p]
print('rc', lb.tgru_debug_stamps_wgrad(out))
print('prologue', out[0], 'stage loop', out[1], '(19 stages:', out[1] / 19, 'per stage) epilogue', out[2], 'cycles')
