import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import timegan_amd as tg
from helpers import load_gru_case, expected
case = load_gru_case('gru_b4_t5_h24_l1'); ref = expected(case, 'f32')
B,T,In,H,L = case['dims']; dev = torch.device('cuda:0')
x = torch.from_numpy(case['x']).to(dev); dy = torch.from_numpy(case['dy']).to(dev)
w = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for layer in case['weights'] for a in layer]
out = tg.gru_forward_backward_raw(x, w, B, T, 1, dy)
dx = out['dx'].cpu().numpy(); err = np.abs(dx - ref['dx'])
np.set_printoptions(linewidth=200, precision=2)
print('err by t:', err.max(axis=(0,2)))
print('err by col:', err.max(axis=(0,1)))
print('err by b:', err.max(axis=(1,2)))
for i,n in enumerate(('dw_ih','dw_hh','db_ih','db_hh')):
    e = np.abs(out['grads'][i].cpu().numpy() - ref['grads'][0][i])
    print(n, 'max err', e.max(), 'by row-block:', [float(e[k*8:(k+1)*8].max()) for k in range(9)])
