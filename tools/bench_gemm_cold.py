"""Dev tool: the QKV / attention-output products with operands and outputs that rotate over buffers far larger than the
256 MiB Infinity Cache (what a layer sees inside the training step: its input was written by another kernel, its output
goes to fresh memory), GEMM settings interleaved in ONE process.  A back-to-back loop over the same buffers (tools/
bench_gemm.py) keeps the operands in the Infinity Cache and ranked a kernel with a shorter DMA lead 6-14 % AHEAD that
is 20-25 % BEHIND here and in the step (round 3, the wave-specialised kernel of commit 10e1a6e)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H = 65536, 768
bf = torch.bfloat16
NB = 6
Xs = [(torch.randn(M, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
Wq = [(torch.randn(3 * H, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
Wo = [(torch.randn(H, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
bq, bo = torch.randn(3 * H, device=dev), torch.randn(H, device=dev)
Oq = [torch.empty(M, 3 * H, dtype=bf, device=dev) for _ in range(NB)]
Oo = [torch.empty(M, H, dtype=bf, device=dev) for _ in range(NB)]
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def run(kind, n=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(n):
        junk.add_(1)                       # 1 GiB of traffic between launches: nothing of the operands stays cached
        e0.record()
        if kind == "qkv": ops.gemm(Xs[i % NB], Wq[i % NB], epilogue="bias", bias=bq, out=Oq[i % NB])
        else: ops.gemm(Xs[i % NB], Wo[i % NB], epilogue="bias", bias=bo, out=Oo[i % NB])
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n
for kind, fl in (("qkv", 2 * M * 2304 * 768), ("out", 2 * M * 768 * 768)):
    res = {}
    for rnd in range(4):
        for vn, cfg in (("nt", 1 + 1024), ("plain", 1 + 2048), ("per-tile nt", 17 + 1024)):
            L.check(lib.czc_gemm_set_tile(cfg))
            t = run(kind)
            if rnd: res.setdefault(vn, []).append(t)
    print(kind, {k: f"{statistics.median(v)*1e3:.1f} us {fl/statistics.median(v)/1e9:.0f} TF/s" for k, v in res.items()}, flush=True)
L.check(lib.czc_gemm_set_tile(0))
