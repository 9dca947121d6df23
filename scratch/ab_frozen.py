"""Which sub-step first makes a trainer without the discarded gradients differ from one that forms them all."""
import sys, torch
sys.path.insert(0, '.')
import timegan_amd as tg
dev = torch.device('cuda:0')
H = int(sys.argv[1]) if len(sys.argv) > 1 else 96
def make(allg):
    cfg = tg.make_cfg(features=5, hidden=H, layers=2, seq_len=8, batch_size=16, device='cuda:0')
    torch.manual_seed(5)
    return tg.TimeGANTrainer(cfg, device=dev, all_gradients=allg)
a, b = make(True), make(False)
g = torch.Generator().manual_seed(6)
x = torch.rand(16, 8, 5, generator=g).to(dev)
noise = [torch.rand(16, 8, 5, generator=g).to(dev) for _ in range(11)]
def diff(tag):
    bad = []
    for k in a.nets:
        for (n, p), (_, q) in zip(a.nets[k].named_parameters(), b.nets[k].named_parameters()):
            if not torch.equal(p, q):
                bad.append((k + '.' + n, float((p - q).abs().max())))
    print(tag, 'differing params:', bad[:6], flush=True)
    return bool(bad)
for rep in range(2):
    for t in (a, b):
        t.supervisor_step(x)
    if diff('sup %d' % rep): break
    for t in (a, b):
        t.embedding_step(x)
    if diff('emb %d' % rep): break
for i in range(4):
    la = a._joint_g_core(x, noise[i]); lb = b._joint_g_core(x, noise[i])
    print('g', i, float(la[0]), float(lb[0]))
    if diff('G %d' % i): break
    la = a._joint_e_core(x); lb = b._joint_e_core(x)
    print('e', i, float(la[0]), float(lb[0]))
    if diff('E %d' % i): break
