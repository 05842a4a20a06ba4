"""Dev tool: weight-gradient GEMM with a token count that is / is not a multiple of the K-tile (64)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
H, I = 768, 3072
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts) * 1e3
for M in (65536, 65504, 33280, 33250):
    YH, Xi = r(M, H), r(M, I)
    t = timeit(lambda: ops.gemm(YH, Xi, a_layout=1, b_layout=1, epilogue="f32", M=H, N=I, K=M, splitk=7))
    print(f"dW2 tokens {M:6d} (tokens % 64 = {M % 64:2d}): {t:7.1f} us  {2 * M * H * I / t / 1e6:7.1f} TF/s", flush=True)
