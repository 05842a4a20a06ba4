"""Passage encoding and retrieval plumbing around the encoder forward and the flat index.

What the reference does in generate_passage_embeddings.py:26-67 and passage_retrieval.py:33-130, arranged for one
MI355X: texts are tokenised slice by slice on the host while the previous slice is still being encoded (nothing on
the path waits for the GPU), every embedding of a run lands in one HBM buffer, and there is ONE device-to-host copy
at the end instead of one per batch.  Shard files keep the reference's format: pickle of `(ids, ndarray [n, H])`."""
from __future__ import annotations

import json
import pickle
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch


def passage_text(p: dict, no_title: bool = False, lowercase: bool = False, normalize: bool = False) -> str:
    """title + " " + text unless told otherwise (generate_passage_embeddings.py:33-40)."""
    text = p["text"] if (no_title or "title" not in p) else p["title"] + " " + p["text"]
    return prepare_text(text, lowercase, normalize)


def prepare_text(text: str, lowercase: bool, normalize: bool) -> str:
    if lowercase:
        text = text.lower()
    if normalize:
        from src import normalize_text
        text = normalize_text.normalize(text)
    return text


def _slices(n: int, size: int) -> Iterator[Tuple[int, int]]:
    for lo in range(0, n, size):
        yield lo, min(n, lo + size)


@torch.no_grad()
def encode_texts(model, tokenizer, texts: Sequence[str], batch_size: int, max_length: int, progress_every: int = 0) -> torch.Tensor:
    """Eval-mode embeddings [len(texts), H] ON THE DEVICE.  Each slice of `batch_size` texts is padded to its own
    longest member and truncated to `max_length` tokens, as `batch_encode_plus(padding=True, truncation=True)` does in
    the reference."""
    call = tokenizer.batch_encode_plus if hasattr(tokenizer, "batch_encode_plus") else tokenizer
    out = None
    for lo, hi in _slices(len(texts), batch_size):
        enc = call(list(texts[lo:hi]), return_tensors="pt", max_length=max_length, padding=True, truncation=True)
        emb = model(**{name: t.cuda(non_blocking=True) for name, t in enc.items()})
        if out is None:
            out = torch.empty(len(texts), emb.shape[1], dtype=emb.dtype, device=emb.device)
        out[lo:hi] = emb
        if progress_every and lo and (lo // batch_size) % max(1, progress_every // batch_size) == 0:
            print(f"Encoded passages {hi}")
    if out is None:
        out = torch.empty(0, model.config.hidden_size, dtype=torch.float32, device="cuda")
    return out


def shard_range(n: int, shard_id: int, num_shards: int) -> Tuple[int, int]:
    """Equal slices, the last shard takes the remainder (generate_passage_embeddings.py:80-84)."""
    size = n // num_shards
    lo = shard_id * size
    return lo, (n if shard_id == num_shards - 1 else lo + size)


def write_shard(path: str, ids: List, embeddings: np.ndarray) -> None:
    with open(path, "wb") as f:
        pickle.dump((ids, embeddings), f)


def read_shard(path: str) -> Tuple[List, np.ndarray]:
    """A shard THIS pipeline (or the reference's generate_passage_embeddings.py) wrote: pickle, so only for own files."""
    with open(path, "rb") as f:
        ids, embeddings = pickle.load(f)
    return list(ids), np.asarray(embeddings)


def fill_index(indexer, shard_paths: Iterable[str], rows_per_add: int) -> None:
    """Add every shard to the index in slices of at most `rows_per_add` rows, in file order; slices may span shard
    boundaries exactly as the reference's accumulate-and-drain loop makes them (passage_retrieval.py:65-91)."""
    held_ids: List = []
    held: List[np.ndarray] = []
    held_rows = 0

    def drain(everything: bool):
        nonlocal held_ids, held, held_rows
        while held_rows > (0 if everything else rows_per_add):
            block = held[0] if len(held) == 1 else np.vstack(held)
            take = min(rows_per_add, block.shape[0])
            indexer.index_data(held_ids[:take], block[:take])
            held_ids, held, held_rows = held_ids[take:], ([block[take:]] if take < block.shape[0] else []), block.shape[0] - take

    for path in shard_paths:
        print(f"Loading file {path}")
        ids, emb = read_shard(path)
        held_ids.extend(ids)
        held.append(emb)
        held_rows += emb.shape[0]
        drain(False)
    drain(True)
    print("Data indexing completed.")


def read_questions(path: str) -> List[dict]:
    with open(path) as f:
        if path.endswith(".jsonl"):
            return [json.loads(line) for line in f]
        if path.endswith(".json"):
            return json.load(f)
    raise ValueError(f"{path}: expected .json or .jsonl")


def attach_contexts(examples: List[dict], passages_by_id: Dict, results) -> None:
    """examples[i]["ctxs"] = the retrieved passages with their scores as strings (passage_retrieval.py:109-127)."""
    if len(examples) != len(results):
        raise ValueError("one result per question expected")
    for ex, (ids, scores) in zip(examples, results):
        ex["ctxs"] = [{"id": pid, "title": passages_by_id[pid]["title"], "text": passages_by_id[pid]["text"], "score": str(sc)}
                      for pid, sc in zip(ids, scores)]


def recall_report(examples: List[dict], workers: int):
    """R@k line of the reference (passage_retrieval.py:94-106) + the per-question hit flags written back as `hasanswer`."""
    from src.evaluation import calculate_matches
    stats = calculate_matches(examples, workers)
    print("Validation results: top k documents hits %s", stats.top_k_hits)
    frac = [v / len(examples) for v in stats.top_k_hits]
    print("".join(f"R@{k}: {frac[k - 1]} " for k in (5, 10, 20, 100) if k <= len(frac)))
    for ex, hits in zip(examples, stats.questions_doc_hits):
        for ctx, hit in zip(ex["ctxs"], hits):
            ctx["hasanswer"] = hit
    return frac
