"""Minimal data path for config 1: jsonl -> tokens -> two random crops -> [CLS]/[SEP]
-> padded batch, producing exactly the batch contract the hot path consumes
(`q_tokens`/`k_tokens` int64 [B, L], `q_mask`/`k_mask` bool [B, L], pad id 0).

Follows the reference's sampling recipe: a chunk of `chunk_length` tokens is cut from a document, the
query and the key are independent `randomcrop`s of it with ratio ~ U[ratio_min, ratio_max]
(src/data.py:553-560), bos/eos are added (src/data.py:635-652) and each side is padded to ITS OWN batch
maximum with id 0 (build_mask, src/data.py:563-573) — q and k generally have different widths.
Not built: LazyDataset byte offsets, the 71 G-token tokens.bin reader, multi-dataset sampling and the
augmentations (CPU-side work outside the MI355X hot path)."""
from __future__ import annotations

import json
import random
from typing import Iterator, List

import torch


def crop(tokens: List[int], ratio_min: float, ratio_max: float, rng: random.Random) -> List[int]:
    ratio = rng.uniform(ratio_min, ratio_max)
    length = int(len(tokens) * ratio)
    start = rng.randint(0, len(tokens) - length)
    return tokens[start:start + length]


def with_bos_eos(tokens: List[int], bos, eos) -> List[int]:
    return ([bos] if bos is not None else []) + list(tokens) + ([eos] if eos is not None else [])


def pad_batch(seqs: List[List[int]]):
    width = max(len(s) for s in seqs)
    ids = torch.zeros(len(seqs), width, dtype=torch.int64)
    mask = torch.zeros(len(seqs), width, dtype=torch.bool)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = torch.tensor(s, dtype=torch.int64)
        mask[i, :len(s)] = True
    return ids, mask


def read_texts(paths) -> List[str]:
    texts = []
    for path in paths:
        with open(path) as f:
            for line in f:
                line = line.strip()
                if line:
                    obj = json.loads(line)
                    texts.append(obj["text"] if isinstance(obj, dict) else str(obj))
    return texts


def loader(opt, tokenizer, rank: int, world: int) -> Iterator:
    if tokenizer is None:
        raise RuntimeError("--train_data needs a local tokenizer (vocab.txt next to the model's config.json)")
    texts = read_texts(opt.train_data)
    docs = [tokenizer.encode(t, add_special_tokens=False) for t in texts]
    docs = [d for d in docs if len(d) >= 8]
    if not docs:
        raise RuntimeError("no usable documents in --train_data")
    rng = random.Random(opt.seed * 7919 + rank)
    bos, eos = tokenizer.bos_token_id, tokenizer.eos_token_id
    B, L = opt.per_gpu_batch_size, opt.chunk_length
    while True:
        qs, ks = [], []
        for _ in range(B):
            d = docs[rng.randrange(len(docs))]
            if len(d) > L:
                s = rng.randint(0, len(d) - L)
                d = d[s:s + L]
            qs.append(with_bos_eos(crop(d, opt.ratio_min, opt.ratio_max, rng), bos, eos))
            ks.append(with_bos_eos(crop(d, opt.ratio_min, opt.ratio_max, rng), bos, eos))
        q_tokens, q_mask = pad_batch(qs)
        k_tokens, k_mask = pad_batch(ks)
        yield {"q_tokens": q_tokens, "q_mask": q_mask, "k_tokens": k_tokens, "k_mask": k_mask}, None
