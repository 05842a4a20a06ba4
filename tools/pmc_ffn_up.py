"""Dev tool: a few FFN-up launches (bias + GELU epilogue, 65536 x 3072 x 768) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
M, H, I = 65536, 768, 3072
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
X, W1 = r(M, H), r(I, H)
b1 = torch.randn(I, device=dev)
for _ in range(4): ops.gemm(X, W1, epilogue="bias_gelu", bias=b1)
torch.cuda.synchronize()
