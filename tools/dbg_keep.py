import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from czech_contriever_amd import lib as L, ops
from tests.dropout_ref import keep_mask
L.load(); dev = torch.device("cuda:0")
B, S, nh, p, seed, site = 2, 128, 2, 0.1, 7, 16
H = nh * 64
qkv = (torch.randn(B * S, 3 * H, device=dev)).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
keep = keep_mask(seed, site, p, B * nh * S * S).reshape(B, nh, S, S)
ctx, lse, words = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site, keep_words=True)
w = words.view(B, nh, 8, 256).cpu().numpy().astype(np.uint32)
got = np.zeros((B, nh, S, S), dtype=np.uint8)
for q in range(S):
    ks = np.arange(S); pos = 32 * (ks // 32) + 8 * ((ks % 32) // 8) + 2 * (ks % 4) + (ks // 4) % 2
    got[:, :, q, :] = (w[:, :, q // 32, pos] >> (q % 32)) & 1
bad = got != keep.astype(np.uint8)
print("mismatches", bad.sum(), "of", bad.size)
idx = np.argwhere(bad)
import collections
print("by b", collections.Counter(idx[:, 0]), "by h", collections.Counter(idx[:, 1]))
print("by q//32", collections.Counter(idx[:, 2] // 32), "by q%32", sorted(collections.Counter(idx[:, 2] % 32).items())[:8])
print("by k//32", collections.Counter(idx[:, 3] // 32), "by k%32", sorted(collections.Counter(idx[:, 3] % 32).items()))
