"""Dev tool: time the LayerNorm row kernels at the bench shape (65536 x 768), hipEvents around repeated launches."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
M, H = 65536, 768
bf = torch.bfloat16
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
X, Y, D, D2 = r(M, H), r(M, H), r(M, H), r(M, H)
gam = torch.ones(H, device=dev); bet = torch.zeros(H, device=dev)
x, st = ops.residual_ln_forward(X, Y, gam, bet, 1e-12, 0.1, 5, 7)
dg, dbb, dbi = (torch.zeros(H, device=dev) for _ in range(3))
def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts) * 1e3
print(f"residual_ln fwd p=0.1        {timeit(lambda: ops.residual_ln_forward(X, Y, gam, bet, 1e-12, 0.1, 5, 7)):7.1f} us")
print(f"residual_ln fwd p=0          {timeit(lambda: ops.residual_ln_forward(X, Y, gam, bet, 1e-12, 0.0, 5, 7)):7.1f} us")
print(f"residual_ln bwd p=0.1        {timeit(lambda: ops.residual_ln_backward(D, None, X, Y, st, gam, dg, dbb, dbi, 0.1, 5, 7)):7.1f} us (+ reduce)")
print(f"residual_ln bwd p=0.1 + dx2  {timeit(lambda: ops.residual_ln_backward(D, D2, X, Y, st, gam, dg, dbb, dbi, 0.1, 5, 7)):7.1f} us (+ reduce)")
print(f"residual_ln bwd p=0          {timeit(lambda: ops.residual_ln_backward(D, None, X, Y, st, gam, dg, dbb, dbi, 0.0, 5, 7)):7.1f} us (+ reduce)")
