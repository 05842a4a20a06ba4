"""numpy restatement of the counter-based dropout RNG in czech_contriever_amd/csrc/czc_common.h
(czc_hash32 / czc_rng_key / czc_rng_quad / czc_keep), so that the oracle can be handed the masks the kernels draw.
Attention sites index element (b, head, q, key) as ((b*heads + head)*S + q) * roundup(S, 4) + key."""
import numpy as np


def _hash32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x.astype(np.uint64) * np.uint64(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def keep_mask(seed: int, site: int, p: float, n: int) -> np.ndarray:
    """uint8[n]: 1 = keep.  Elements 4i..4i+3 use the four 16-bit fields of one 64-bit draw."""
    thr = np.uint32(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))
    key = _hash32(np.array([(seed + 0x9E3779B9 * (site + 1)) & 0xFFFFFFFF], dtype=np.uint64).astype(np.uint32))[0]
    e = np.arange(n, dtype=np.uint32)
    x = (e >> np.uint32(2)) ^ key
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    lo = (x.astype(np.uint64) * np.uint64(0x846CA68B)).astype(np.uint32)
    lo ^= lo >> np.uint32(16)
    hi = (x.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
    hi ^= hi >> np.uint32(16)
    w = np.where(e & np.uint32(2), hi, lo)
    d = np.where(e & np.uint32(1), w >> np.uint32(16), w & np.uint32(0xFFFF))
    return (d >= thr).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# the assembled encoder's dropout sites (czech_contriever_amd/csrc/encoder.hip: SITE_EMB, SITE_ATTN(l), SITE_H1(l),
# SITE_H2(l); split-range shift +4) and the engine's per-forward seed (czech_contriever_amd/engine.py:_forward_impl)
# ------------------------------------------------------------------------------------------------
SITE_EMB = 1


def site_attn(l):
    return 16 * (l + 1) + 0


def site_h1(l):
    return 16 * (l + 1) + 1


def site_h2(l):
    return 16 * (l + 1) + 2


SPLIT_SHIFT = 4


def engine_seed(initial_seed: int, drop_calls: int, salt: int) -> int:
    """Seed of an encoder's `drop_calls`-th TRAINING forward (counted from 1) in a process whose torch.initial_seed() is
    `initial_seed`; salt = the encoder's role (MoCo: query 0, key 1)."""
    return (initial_seed * 2654435761 + drop_calls * 40503 + (salt + 1) * 0x9E3779B1) & 0xFFFFFFFF


def _scale(keep, p):
    import torch
    return torch.from_numpy(keep.astype(np.float32)) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def rows_scale(seed, site, p, rows, H):
    """[rows, H] keep / (1 - p) of a row kernel launched on `rows` rows: element (r, c) is draw r * H + c."""
    return _scale(keep_mask(seed, site, p, rows * H).reshape(rows, H), p)


def attn_scale(seed, site, p, B, nh, S):
    """[B, nh, S, S] keep / (1 - p) of an attention launch on B sequences: the draw index pitch is S rounded up to 4."""
    S4 = (S + 3) // 4 * 4
    return _scale(keep_mask(seed, site, p, B * nh * S * S4).reshape(B, nh, S, S4)[..., :S], p)


def encoder_scales(seed, cfg, B, S, split=0):
    """Every dropout site of one training forward of the encoder as multiplicative scales (keep / (1 - p)), keyed as
    oracle.encoder_forward(masks=...) wants them: "emb", ("attn", l), ("h1", l), ("h2", l).  split > 0: the last layer ran
    in the batch ranges [0, split) and [split, B) (czc_encoder_forward's split_batch): each range draws from index 0 of its
    own launch, the second with its sites shifted by SPLIT_SHIFT."""
    import torch
    H, nh, L = cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers
    ph, pa = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    out = {}
    if ph > 0:
        out["emb"] = rows_scale(seed, SITE_EMB, ph, B * S, H).view(B, S, H)
    for l in range(L):
        ranges = [(0, B, 0)] if not (split and l == L - 1) else [(0, split, 0), (split, B - split, SPLIT_SHIFT)]
        if pa > 0:
            out[("attn", l)] = torch.cat([attn_scale(seed, site_attn(l) + sh, pa, nb, nh, S) for _, nb, sh in ranges])
        if ph > 0:
            out[("h1", l)] = torch.cat([rows_scale(seed, site_h1(l) + sh, ph, nb * S, H).view(nb, S, H) for _, nb, sh in ranges])
            out[("h2", l)] = torch.cat([rows_scale(seed, site_h2(l) + sh, ph, nb * S, H).view(nb, S, H) for _, nb, sh in ranges])
    return out


class DeviceScales:
    """encoder_scales() for shapes whose masks are too many for numpy (B 256 x 12 heads x 256 x 256 per layer): the same
    dict interface (`get`), each site drawn on demand by czc_dropout_mask -- the kernels' own czc_keep, equal to keep_mask
    above element for element (tests/test_gpu_dropout_parity.py checks that) -- and copied to the host."""

    def __init__(self, seed, cfg, B, S, device):
        self.seed, self.cfg, self.B, self.S, self.device = seed, cfg, B, S, device

    def get(self, key, default=None):
        import torch
        from czech_contriever_amd import ops
        cfg, B, S = self.cfg, self.B, self.S
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        if key == "emb":
            site, p, shape = SITE_EMB, cfg.hidden_dropout_prob, (B, S, H)
        else:
            kind, l = key
            site = {"attn": site_attn, "h1": site_h1, "h2": site_h2}[kind](l)
            p = cfg.attention_probs_dropout_prob if kind == "attn" else cfg.hidden_dropout_prob
            shape = (B, nh, S, (S + 3) // 4 * 4) if kind == "attn" else (B, S, H)
        if p <= 0:
            return default
        n = 1
        for d in shape:
            n *= d
        keep = ops.dropout_mask(self.seed, site, p, n, self.device).view(shape)
        if key != "emb" and key[0] == "attn":
            keep = keep[..., :S]
        return keep.cpu().to(torch.float32) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
