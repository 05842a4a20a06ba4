"""Dev tool: the same 65536 x 3072 x 768 product under each bf16 epilogue (what the epilogue itself costs)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
X, W1, YI = r(M, H), r(I, H), r(M, I)
b1 = torch.randn(I, device=dev)
def gelu1():
    ops.gemm(X, W1, epilogue="bias_gelu_only", bias=b1)
cases = [("store", lambda: ops.gemm(X, W1)), ("store again", lambda: ops.gemm(X, W1)), ("bias", lambda: ops.gemm(X, W1, epilogue="bias", bias=b1)),
         ("bias+gelu (2 out)", lambda: ops.gemm(X, W1, epilogue="bias_gelu", bias=b1)), ("bias+gelu (1 out)", gelu1),
         ("add", lambda: ops.gemm(X, W1, epilogue="add", aux=YI)), ("dgelu", lambda: ops.gemm(X, W1, epilogue="dgelu", aux=YI))]
def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(6):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts[1:]) * 1e3
for name, fn in cases:
    t = timeit(fn)
    print(f"{name:20s} {t:7.1f} us  {2*M*H*I/t/1e6:7.1f} TF/s", flush=True)
