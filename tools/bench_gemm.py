"""Dev tool: time the GEMM shapes of one bert-base layer (B*S = 65536 tokens) under both tile configurations."""
import os, sys
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
]
def dw(name, dy, x, No, Ko):
    for sp in (None,):
        pass
    return name
def splits_for(Mo, No, tile):
    tiles = -(-Mo // tile) * -(-No // tile)
    return max(1, min(1024, (256 if tile == 256 else 512) // tiles))
dws = [("dW2 [768,3072]", YH, Xi, 768, 3072), ("dW1 [3072,768]", YI, X, 3072, 768), ("dWo [768,768]", YH, X, 768, 768),
       ("dWqkv [2304,768]", Y3, X, 2304, 768)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, tile in ((1, 256), (2, 128)):
    L.check(lib.czc_gemm_set_tile(cfg))
    tot = 0.0
    for name, fn, fl in cases:
        ms = timeit(fn); tot += ms
        print(f"cfg{cfg} {name:36s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
    for name, dy, x, Mo, No in dws:
        for mult in (1, 2):
            sp = splits_for(Mo, No, tile) * mult
            fn = lambda: ops.gemm(dy, x, a_layout=1, b_layout=1, epilogue="f32", M=Mo, N=No, K=M, splitk=sp)
            ms = timeit(fn)
            print(f"cfg{cfg} {name:20s} splitk={sp:3d}            {ms*1e3:8.1f} us  {2*M*Mo*No/ms/1e9:7.1f} TF/s", flush=True)
    print(f"cfg{cfg} sum of 8 activation GEMMs: {tot*1e3:.0f} us", flush=True)
L.check(lib.czc_gemm_set_tile(0))
