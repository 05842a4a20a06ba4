"""Dev tool: the embedding + LayerNorm backward over token ids of different shapes (uniform ids as bench.py draws them;
text-like ids: [CLS] first, Zipf-distributed words, a padded tail of id 0), hipEvents."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
L.load(); dev = torch.device("cuda:0")
B, S, H, V = 256, 256, 768, 51200
g = torch.Generator(device="cpu").manual_seed(0)
word = torch.randn(V, H, device=dev) * 0.02; pos = torch.randn(512, H, device=dev) * 0.02; type0 = torch.randn(H, device=dev) * 0.02
gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
dx0 = (torch.randn(B * S, H, device=dev) * 0.1).to(torch.bfloat16)
def timeit(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)
def ids_uniform():
    return torch.randint(1000, V, (B, S), generator=g)
def ids_text(pad_frac):
    r = torch.rand(B, S, generator=g)
    ids = (1000 + ((V - 1000) * r.pow(6.0)).long()).clamp(max=V - 1)   # heavy head: a few hundred ids take most draws
    ids[:, 0] = 101
    lens = (S * (1.0 - pad_frac * torch.rand(B, generator=g) * 2).clamp(min=0.1)).long()
    for b in range(B):
        ids[b, lens[b] - 1] = 102
        ids[b, lens[b]:] = 0
    return ids
for name, ids in [("uniform ids", ids_uniform()), ("text-like, no padding", ids_text(0.0)), ("text-like, ~30 % padding", ids_text(0.3))]:
    ids = ids.to(dev)
    dx = (dx0.view(B, S, H) * (ids != 0)[..., None]).view(B * S, H).contiguous()   # padded positions receive an exactly-zero gradient
    x, stats = ops.embed_ln_forward(ids, word, pos, type0, gamma, beta, 1e-12)
    d_word = torch.zeros_like(word); d_pos = torch.zeros_like(pos); d_type0 = torch.zeros_like(type0)
    d_gamma = torch.zeros_like(gamma); d_beta = torch.zeros_like(beta)
    ms = timeit(lambda: ops.embed_ln_backward(ids, word, pos, type0, gamma, stats, dx, d_word, d_pos, d_type0, d_gamma, d_beta))
    uniq = int(torch.unique(ids).numel())
    print(f"{name:28s} {uniq:6d} distinct ids, most frequent {int(torch.bincount(ids.flatten()).max()):6d} times: {ms*1e3:8.1f} us")
