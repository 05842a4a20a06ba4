"""Dev tool: A/B the GEMM shapes of one bert-base layer (65536 tokens) across settings, interleaved in ONE
process (variants x rounds), reporting the median per variant (cdna guide §5.4 rule 24)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
bf = torch.bfloat16
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
X, Xi = r(M, H), r(M, I)
W = {"qkv": r(3 * H, H), "o": r(H, H), "w1": r(I, H), "w2": r(H, I)}
Y3, YH, YI = r(M, 3 * H), r(M, H), r(M, I)
bias = {k: torch.randn(n, device=dev) for k, n in (("qkv", 3 * H), ("o", H), ("w1", I), ("w2", H))}
cases = [
    ("fwd qkv   bias      M,2304,768", lambda: ops.gemm(X, W["qkv"], epilogue="bias", bias=bias["qkv"]), 2 * M * 2304 * 768),
    ("fwd out   bias      M,768,768", lambda: ops.gemm(X, W["o"], epilogue="bias", bias=bias["o"]), 2 * M * 768 * 768),
    ("fwd ffn1  bias_gelu M,3072,768", lambda: ops.gemm(X, W["w1"], epilogue="bias_gelu", bias=bias["w1"]), 2 * M * 3072 * 768),
    ("fwd ffn2  bias      M,768,3072", lambda: ops.gemm(Xi, W["w2"], epilogue="bias", bias=bias["w2"]), 2 * M * 768 * 3072),
    ("bwd dH    dgelu     M,3072,768", lambda: ops.gemm(YH, W["w2"], b_layout=1, epilogue="dgelu", aux=YI), 2 * M * 3072 * 768),
    ("bwd dX1   add       M,768,3072", lambda: ops.gemm(YI, W["w1"], b_layout=1, epilogue="add", aux=YH), 2 * M * 768 * 3072),
    ("bwd dctx  store     M,768,768", lambda: ops.gemm(YH, W["o"], b_layout=1), 2 * M * 768 * 768),
    ("bwd dXin  add       M,768,2304", lambda: ops.gemm(Y3, W["qkv"], b_layout=1, epilogue="add", aux=YH), 2 * M * 768 * 2304),
    ("dW2 [768,3072] splitk 7", lambda: ops.gemm(YH, Xi, a_layout=1, b_layout=1, epilogue="f32", M=768, N=3072, K=M, splitk=7), 2 * M * 768 * 3072),
    ("dWqkv [2304,768] splitk 9", lambda: ops.gemm(Y3, X, a_layout=1, b_layout=1, epilogue="f32", M=2304, N=768, K=M, splitk=9), 2 * M * 768 * 2304),
]
variants = [("tile256", 1, 0), ("tile128", 2, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "pp":
    variants = [("plain", 65, 0), ("pingpong", 33, 0), ("auto", 1, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "persist":
    variants = [("persist", 1, 0), ("per-tile", 17, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "stores":
    variants = [("plain", 1 + 2048, 0), ("nt", 1 + 1024, 0), ("sc1", 1 + 512, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "tiles":
    variants = [("tile256", 1, 0), ("tile128", 2, 0)]
def timeit(fn, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = {v[0]: 0.0 for v in variants}
for name, fn, fl in cases:
    res = {v[0]: [] for v in variants}
    for rnd in range(6):
        for vn, cfg, stag in variants:
            L.check(lib.czc_gemm_set_tile(cfg))
            t = timeit(fn)
            if rnd: res[vn].append(t)
    line = f"{name:32s}"
    for vn, _, _ in variants:
        med = statistics.median(res[vn]); tot[vn] += med
        line += f" | {vn}: {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF/s"
    print(line, flush=True)
print("sums:", {k: round(v * 1e3) for k, v in tot.items()})
L.check(lib.czc_gemm_set_tile(0))
