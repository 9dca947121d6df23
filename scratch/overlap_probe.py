"""Does a batched GEMM on a second stream fill the gaps of the per-step recurrent kernels?
Runs (a) one C4 layer forward (input projection + 256 step launches), (b) N input-projection GEMMs of the
C4 shape, alone and then concurrently on two streams, and prints wall times."""
import sys
import time
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timegan_amd as tg
from bench import make_weights

dev = torch.device("cuda:0")
B, T, H = 2048, 256, 512
x = torch.rand(T, B, H, device=dev)
runner = tg.GRURunner(B, T, H, H, 1, tg.LAYOUT_TBF, dev)
runner.set_weights(make_weights(H, H, 1, dev, 42))
a = torch.rand(T * B // 2, H, device=dev)
w = torch.rand(3 * H, H, device=dev)
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
print("streams", sa.cuda_stream, sb.cuda_stream)
NG = 4


def run_a():
    with torch.cuda.stream(sa):
        runner.forward(x)


def run_b():
    with torch.cuda.stream(sb):
        for _ in range(NG):
            tg.input_projection(a, w)


def timed(fns, n=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns:
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ta = timed([run_a])
tb = timed([run_b])
tab = timed([run_a, run_b])
tba = timed([run_b, run_a])
print(f"layer forward alone {ta:.2f} ms; {NG} half-size GEMMs alone {tb:.2f} ms; both streams {tab:.2f} / {tba:.2f} ms "
      f"(sum {ta + tb:.2f})")
