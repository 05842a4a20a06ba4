#!/usr/bin/env python3
"""Dev helper: time the flat-index search kernels. usage: bench_index.py [rows] [nq] [k] [dim] [coarse|exact]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd.flat_index import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 768
mode = sys.argv[5] if len(sys.argv) > 5 else "coarse"
g = torch.Generator(device="cuda").manual_seed(0)
ix = FlatIPIndex(dim)
blk = 1_000_000
for s in range(0, rows, blk):
    ix.add(torch.randn(min(blk, rows - s), dim, device="cuda", generator=g) * 0.05)
q = torch.randn(nq, dim, device="cuda", generator=g)
ix.search(q, k, mode=mode); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps):
    s, r = ix.search(q, k, mode=mode)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * rows * nq * dim
print(f"rows={rows} nq={nq} k={k} dim={dim} mode={mode}: {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s of scoring (fp32 matrix peak 157.3, bf16 2500)  {nq / ms * 1e3:.0f} queries/s  db pass {rows * dim * 4 / ms / 1e6:.0f} GB/s")
