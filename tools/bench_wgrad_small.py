"""Dev tool: the 768 x 768 weight gradient (attention output projection: 9 tiles of 256 x 256, so 28-way split-K and 28
slabs to sum) under other tile / split choices, slab reduction included, interleaved in one process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H = 65536, 768
bf = torch.bfloat16
NB = 4
Y = [(torch.randn(M, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
X = [(torch.randn(M, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
dst = torch.zeros(H, H, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
fl = 2 * M * H * H
def run(cfg, sp, n=8):
    L.check(lib.czc_gemm_set_tile(cfg))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(n):
        junk.add_(1)
        e0.record()
        slabs = ops.gemm(Y[i % NB], X[i % NB], a_layout=1, b_layout=1, epilogue="f32", M=H, N=H, K=M, splitk=sp)
        ops.reduce_slabs(slabs, dst, False)
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    L.check(lib.czc_gemm_set_tile(0))
    return tot / n
variants = [("256 tile x28", 1, 28), ("256 tile x14", 1, 14), ("128 tile x14", 2, 14), ("128 tile x7", 2, 7), ("128 tile x10", 2, 10)]
res = {}
for rnd in range(4):
    for name, cfg, sp in variants:
        t = run(cfg, sp)
        if rnd: res.setdefault(name, []).append(t)
for name, _, _ in variants:
    med = statistics.median(res[name])
    print(f"{name:16s} {med*1e3:7.1f} us (gemm + slab sum) {fl/med/1e9:7.1f} TF/s", flush=True)
