"""Dev tool: weight-gradient GEMM (both operands MN-major, split-K f32 slabs) against the same product with K-major
operands, interleaved in one process — separates the cost of the transposing-read path from the split-K machinery."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
bf = torch.bfloat16
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
YH, Xi = r(M, H), r(M, I)
YHt, Xit = r(H, M), r(I, M)
fl = 2 * M * H * I
W1, W1t = r(I, H), r(H, I)
cases = [(f"wgrad A mn B mn splitk 7", lambda: ops.gemm(YH, Xi, a_layout=1, b_layout=1, epilogue="f32", M=H, N=I, K=M, splitk=7)),
         (f"wgrad A k  B mn splitk 7", lambda: ops.gemm(YHt, Xi, a_layout=0, b_layout=1, epilogue="f32", splitk=7)),
         (f"wgrad A k  B k  splitk 7", lambda: ops.gemm(YHt, Xit, epilogue="f32", splitk=7)),
         ("dX1 B mn store", lambda: ops.gemm(Xi, W1, b_layout=1)),
         ("dX1 B mn add", lambda: ops.gemm(Xi, W1, b_layout=1, epilogue="add", aux=YH)),
         ("dX1 B k  store", lambda: ops.gemm(Xi, W1t)),
         ("dX1 B k  add", lambda: ops.gemm(Xi, W1t, epilogue="add", aux=YH)),
         ("dX1 B k  bias", lambda: ops.gemm(Xi, W1t, epilogue="bias", bias=torch.zeros(H, device=dev)))]
def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {c[0]: [] for c in cases}
for rnd in range(6):
    for name, fn in cases:
        t = timeit(fn)
        if rnd: res[name].append(t)
for name, _ in cases:
    med = statistics.median(res[name])
    print(f"{name:22s} {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF/s", flush=True)
