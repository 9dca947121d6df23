import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import timegan_amd as tg
from timegan_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libtgru.so', 'libtgru_stamps.so')
dev = torch.device('cuda:0')
B,T,In,H,L = 128,96,24,24,3
w=[]
for l in range(L):
    for s in ((3*H, In if l==0 else H),(3*H,H),(3*H,),(3*H,)): w.append((torch.rand(s, device=dev)*2-1)/H**0.5)
x = torch.rand(T,B,In, device=dev); dy = torch.randn(T,B,H, device=dev)
r = tg.GRURunner(B,T,In,H,L,tg.LAYOUT_TBF,dev); r.set_weights(w)
for _ in range(50): r.forward(x); r.backward(dy)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong*128)()
_lib.lib().tgru_debug_stamps_vec.argtypes=[ctypes.c_void_p]; print('rc', _lib.lib().tgru_debug_stamps_vec(out))
a = np.array(out[:]).reshape(16,8)[[0,1,2,8,9,10],:6]
print('per wave cycles: [0 reads+flush, 1 fma, 2 gates, 3 bookkeeping, 4 barrier, 5 loop-top]'); np.set_printoptions(suppress=True, linewidth=200); print('fwd waves 0-2, bwd waves 0-2; per tick:'); print((a/98).round(0)); print('sum', (a.sum(1)/98).round(0))
