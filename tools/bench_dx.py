"""Dev tool: the input-gradient products in their K-major form (transposed weight mirror): plain / ping-pong / auto."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
YH, YI, Y3 = r(M, H), r(M, I), r(M, 3 * H)
W2t, W1t, Wot, Wqt = r(I, H), r(H, I), r(H, H), r(H, 3 * H)
cases = [("dH   dgelu M,3072,768", lambda: ops.gemm(YH, W2t, epilogue="dgelu", aux=YI), 2 * M * I * H),
         ("dX1  add   M,768,3072", lambda: ops.gemm(YI, W1t, epilogue="add", aux=YH), 2 * M * I * H),
         ("dctx store M,768,768", lambda: ops.gemm(YH, Wot), 2 * M * H * H),
         ("dXin add   M,768,2304", lambda: ops.gemm(Y3, Wqt, epilogue="add", aux=YH), 2 * M * H * 3 * H)]
variants = [("plain", 65), ("pingpong", 33), ("auto", 1)]
def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, fn, fl in cases:
    res = {v[0]: [] for v in variants}
    for rnd in range(6):
        for vn, cfg in variants:
            L.check(lib.czc_gemm_set_tile(cfg))
            t = timeit(fn)
            if rnd: res[vn].append(t)
    line = f"{name:24s}"
    for vn, _ in variants:
        med = statistics.median(res[vn]); line += f" | {vn}: {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF/s"
    print(line, flush=True)
L.check(lib.czc_gemm_set_tile(0))
