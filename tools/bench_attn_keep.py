"""Dev tool: the attention backward as a saved encoder pass runs it (keep words from the forward), hipEvents."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load(); dev = torch.device("cuda:0")
B, S, nh, H = 256, 256, 12, 768
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B * S, H, device=dev) * 0.5).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
ctx, lse, words = ops.attention_forward(qkv, mask, B, S, nh, 0.1, 5, 3, keep_words=True)
db = torch.zeros(3 * H, device=dev)
def timeit(fn, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ts = [timeit(lambda: ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, 0.1, 5, 3, d_bias=db, keep_words=words)) for _ in range(6)][1:]
print(f"bwd keep {statistics.median(ts)*1e3:.1f} us (+ partial buffer allocation and reduction)")
