"""Dev tool: attention forward/backward timing with and without dropout (interleaved)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load()
dev = torch.device("cuda:0")
B, S, nh, H = 256, 256, 12, 768
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B * S, H, device=dev) * 0.5).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
mask2 = mask.clone(); mask2[:, 200:] = False
def timeit(fn, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
for rnd in range(5):
    for name, p, mk in (("drop0.1", 0.1, mask), ("nodrop", 0.0, mask), ("drop0.1 padded", 0.1, mask2)):
        ctx, lse = ops.attention_forward(qkv, mk, B, S, nh, p, 5, 3)
        db = torch.zeros(3 * H, device=dev)
        tf = timeit(lambda: ops.attention_forward(qkv, mk, B, S, nh, p, 5, 3))
        if p > 0 and rnd:
            res.setdefault(name + " fwd+keep words", []).append((timeit(lambda: ops.attention_forward(qkv, mk, B, S, nh, p, 5, 3, keep_words=True)), 0.0))
        tb = timeit(lambda: ops.attention_backward(qkv, mk, ctx, dctx, lse, B, S, nh, p, 5, 3, d_bias=db))
        if rnd: res.setdefault(name, []).append((tf, tb))
for k, v in res.items():
    print(f"{k:16s} fwd {statistics.median(x[0] for x in v)*1e3:7.1f} us   bwd (dq+dkdv+reduce) {statistics.median(x[1] for x in v)*1e3:7.1f} us")
