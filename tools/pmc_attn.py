"""Dev tool: a few attention forward / backward launches at the bench shape (B 256, S 256, 12 heads, dropout 0.1) for
rocprofv3 --pmc passes (put `python3 tools/pmc_attn.py` after `--`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
B, S, nh, H = 256, 256, 12, 768
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B * S, H, device=dev) * 0.5).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
db = torch.zeros(3 * H, device=dev)
for _ in range(4):   # as a saved encoder pass runs them: the forward keeps its dropout decisions for the backward
    ctx, lse, words = ops.attention_forward(qkv, mask, B, S, nh, 0.1, 5, 3, keep_words=True)
    ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, 0.1, 5, 3, d_bias=db, keep_words=words)
for _ in range(4):   # and as a pass nothing differentiates through (the key encoder)
    ops.attention_forward(qkv, mask, B, S, nh, 0.1, 5, 3)
torch.cuda.synchronize()
