"""Dev tool: a handful of launches of one GEMM variant (argv[1]) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
bf = torch.bfloat16
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
Xi, W1, W1t = r(M, I), r(I, H), r(H, I)
which = sys.argv[1]
for _ in range(3):
    if which == "mn": ops.gemm(Xi, W1, b_layout=1)
    else: ops.gemm(Xi, W1t)
torch.cuda.synchronize()
