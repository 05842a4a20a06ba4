"""Dev tool: ping-pong K-loop vs plain K-loop, bit for bit, over many shapes and repeats (a RAW / WAR slip in the
half-tile schedule would show as a rare mismatch)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
bf = torch.bfloat16
rng = random.Random(1)
shapes = [(65536, 3072, 768), (65536, 768, 3072), (65536, 2304, 768), (16600, 1280, 192), (33280, 768, 768), (4096, 4096, 4096),
          (8192, 1024, 8192), (1000, 2304, 768), (520, 768, 256), (65536, 768, 768)]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    M, N, K = shapes[it % len(shapes)]
    al, bl = rng.choice([(0, 0), (0, 0), (0, 1), (1, 1)])
    if al == 1: M, N, K = 768, 3072 if it % 2 else 2304, 65536 if it % 3 else 33280
    g = torch.Generator(device=dev).manual_seed(it)
    A = (torch.randn((M, K) if al == 0 else (K, M), device=dev, generator=g) * 0.5).to(bf)
    B = (torch.randn((N, K) if bl == 0 else (K, N), device=dev, generator=g) * 0.5).to(bf)
    kw = dict(a_layout=al, b_layout=bl, M=M, N=N, K=K)
    if al == 1: kw.update(epilogue="f32", splitk=7)
    outs = []
    for cfg in (65, 33, 33):
        L.check(lib.czc_gemm_set_tile(cfg))
        outs.append(ops.gemm(A, B, **kw).clone())
    L.check(lib.czc_gemm_set_tile(0))
    ok = torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    if not ok:
        bad += 1
        print("MISMATCH", it, (al, bl), (M, N, K), float((outs[0].float() - outs[1].float()).abs().max()), flush=True)
print("iterations done, mismatches:", bad, flush=True)
