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
