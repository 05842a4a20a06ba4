"""Dev tool: random attention problems (batch, length, heads, ragged padding, dropout on / off, keep words or regenerated
draws) through the C ABI against an fp32 torch reference with autograd -- forward, dQ / dK / dV, the QKV bias column
sums, and run-to-run bit equality of the backward.  The parametrised tests pin the kernel-selection boundaries (32-query
blocks, 256 tokens, more items than CUs); this walks the space between them.
python tools/fuzz_attention.py [cases] [seed]
tests/test_gpu_fullsize.py runs a seeded 40-case slice of it (`run(40, seed)`) inside pytest -m gpu."""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from czech_contriever_amd import ops
from tests.dropout_ref import keep_mask

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def ref_attn(qkv, mask, B, S, nh, keep, p):
    x = qkv.view(B, S, 3, nh, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    s = s.masked_fill(~mask[:, None, None, :], torch.finfo(torch.float32).min)
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B * S, nh * 64)


def excess(got, ref, atol, rtol):
    return float(((got.float() - ref).abs() - rtol * ref.abs()).max()) - atol


def run(n_cases=120, seed=0, verbose=True):
    """-> (cases, [(tag, why)])"""
    rng = random.Random(seed)
    g = torch.Generator(device="cpu").manual_seed(rng.randrange(1 << 30))
    fails = []
    for case in range(n_cases):
        kind = rng.random()
        if kind < 0.3:
            S = rng.randrange(1, 65)
        elif kind < 0.75:
            S = rng.choice([31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 160, 191, 192, 224, 252, 255, 256]) if rng.random() < 0.6 else rng.randrange(65, 257)
        else:
            S = rng.randrange(257, 513)
        nh = rng.choice([1, 2, 3, 4, 5, 8, 12])
        budget = 24_000_000 // (nh * S * S)                       # elements of the probability tensors of the reference
        B = max(1, min(rng.choice([1, 2, 3, 7, 24, 40, 103]), budget))
        p = 0.1 if (S % 4 == 0 and rng.random() < 0.5) else 0.0
        use_words = p > 0 and S <= 256 and rng.random() < 0.6
        seed, site = rng.randrange(1 << 20), rng.randrange(1, 400)
        H = nh * 64
        qkv = (torch.randn(B * S, 3 * H, generator=g) * rng.choice([0.5, 1.0, 2.0])).to(dev).to(BF16)
        lengths = [S if rng.random() < 0.3 else rng.randrange(1, S + 1) for _ in range(B)]
        mask = torch.zeros(B, S, dtype=torch.bool, device=dev)
        for b, n in enumerate(lengths):
            mask[b, :n] = True
        tag = f"#{case} B{B} S{S} nh{nh} p{p} words{int(use_words)} seed{seed} site{site}"
        keep = None
        if p > 0:
            keep = torch.from_numpy(keep_mask(seed, site, p, B * nh * S * S).reshape(B, nh, S, S)).to(dev).float()
        words = None
        if use_words:
            ctx, lse, words = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site, keep_words=True)
        else:
            ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site)
        qv = qkv.float().requires_grad_(True)
        ref = ref_attn(qv, mask, B, S, nh, keep, p)
        dctx = torch.randn(B * S, H, generator=g).to(dev).to(BF16)
        ref.backward(dctx.float())
        dbias = torch.zeros(3 * H, device=dev)
        dqkv = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site, d_bias=dbias, keep_words=words)
        again = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site, keep_words=words)
        torch.cuda.synchronize()
        scale = float(qkv.float().abs().max())
        why = []
        # rows of padded queries are defined (the reference attends over the valid keys for them too): compare everything
        if not torch.isfinite(ctx.float()).all() or not torch.isfinite(dqkv.float()).all():
            why.append("non-finite output")
        else:
            e = excess(ctx, ref.detach(), 2e-2 * scale, 1e-2)
            if e > 0:
                why.append(f"fwd excess {e:.4f}")
            e = excess(dqkv, qv.grad, 5e-2 * max(1.0, scale), 2e-2)
            if e > 0:
                why.append(f"dqkv excess {e:.4f}")
            e = excess(dbias, qv.grad.sum(0), 2e-2 * math.sqrt(B * S) * max(1.0, scale), 2e-2)
            if e > 0:
                why.append(f"d_bias excess {e:.4f}")
            if not torch.equal(dqkv, again):
                why.append("backward not bit-reproducible")
        if why:
            fails.append((tag, "; ".join(why)))
        if verbose and case % 20 == 19:
            print(f"{case + 1} cases, {len(fails)} failures", flush=True)
    return n_cases, fails


if __name__ == "__main__":
    n, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for t, why in fails:
        print("FAIL", t, why)
    print(f"fuzz_attention: {n} cases, {len(fails)} failures")
    sys.exit(1 if fails else 0)
