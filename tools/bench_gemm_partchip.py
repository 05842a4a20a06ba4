"""Dev tool: is a K = 768 product's per-tile time set by the CU or by the chip?  The same tiles per workgroup (9 / 12) on
all 256 CUs and, with a diagnostic build whose persistent grid is 128 or 64 workgroups (-DCZC_SLOTS_OVERRIDE), on part of
the chip at proportionally fewer rows: cold operands (buffers rotate, 1 GiB of traffic between launches).
python tools/bench_gemm_partchip.py ROWS"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
M, H = int(sys.argv[1]), 768
bf = torch.bfloat16
NB = 4
Xs = [(torch.randn(M, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
Wq = [(torch.randn(3 * H, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
W1 = [(torch.randn(4 * H, H, device=dev) * 0.5).to(bf) for _ in range(NB)]
bq, b1 = torch.randn(3 * H, device=dev), torch.randn(4 * H, device=dev)
Oq = [torch.empty(M, 3 * H, dtype=bf, device=dev) for _ in range(NB)]
O1 = [torch.empty(M, 4 * H, dtype=bf, device=dev) for _ in range(NB)]
O2 = [torch.empty(M, 4 * H, dtype=bf, device=dev) for _ in range(NB)]
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def run(kind, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(n):
        junk.add_(1)
        e0.record()
        j = i % NB
        if kind == "qkv": ops.gemm(Xs[j], Wq[j], epilogue="bias", bias=bq, out=Oq[j])
        elif kind == "ffn_up2": ops.gemm(Xs[j], W1[j], epilogue="bias_gelu", bias=b1, out=O1[j], out2=O2[j])
        else: ops.gemm(Xs[j], W1[j], epilogue="bias_gelu_only", bias=b1, out=O1[j])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
for kind, n in (("qkv", 3 * H), ("ffn_up2", 4 * H), ("ffn_up1", 4 * H)):
    t = run(kind)
    print(f"rows {M} variant {os.environ.get('CZC_LIB_VARIANT', 'base')} {kind}: {t*1e3:.1f} us  {2*M*n*H/t/1e9:.0f} TF/s", flush=True)
