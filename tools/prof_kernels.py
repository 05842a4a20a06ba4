"""Dev tool: launch the hot kernels a few times at full size (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
B, S, nh, H, I = 256, 256, 12, 768, 3072
M = B * S
bf = torch.bfloat16
def r(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
qkv = r(M, 3 * H); mask = torch.ones(B, S, dtype=torch.bool, device=dev)
dctx = r(M, H)
X = r(M, H); W1 = r(I, H); b1 = torch.randn(I, device=dev)
gam = torch.ones(H, device=dev); bet = torch.zeros(H, device=dev)
for _ in range(3):
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, 0.1, 5, 3)
    db = torch.zeros(3 * H, device=dev)
    ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, 0.1, 5, 3, d_bias=db)
    ops.gemm(X, W1, epilogue="bias_gelu", bias=b1)
    x, st = ops.residual_ln_forward(X, dctx, gam, bet, 1e-12, 0.1, 5, 7)
    dg, dbb, dbi = (torch.zeros(H, device=dev) for _ in range(3))
    ops.residual_ln_backward(X, None, X, dctx, st, gam, dg, dbb, dbi, 0.1, 5, 7)
torch.cuda.synchronize()
