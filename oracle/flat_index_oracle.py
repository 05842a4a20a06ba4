"""CPU oracle for the flat inner-product passage search (TEST INFRASTRUCTURE ONLY).

Restates what `faiss.IndexFlatIP.search` returns for the call at Ajchler/czech-contriever src/index.py:43
(passage_retrieval.py:184-186): for every query the k largest inner products with the indexed vectors, descending,
with the matching row numbers; a result shorter than k is padded with row -1 and score -FLT_MAX.  Equal scores are
put in ascending row order (faiss leaves their order unspecified).

faiss (environment.yml pins faiss-gpu 1.7.4) is not installed in this image and is not part of /root/reference:
the algorithm lives in a third-party dependency, so this oracle is a restatement of its published definition
(exact inner product, flat scan).  The reference holds no fixture for it: PARITY UNPINNED by the reference, pinned
only by definition — the scores are computed in float64 here, and on integer-valued vectors every partial sum is
exact in fp32 as well, which is what the bit-exact tests use.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np

FLT_LOWEST = np.float32(-3.4028234663852886e38)


def search(db: np.ndarray, queries: np.ndarray, k: int, dtype=np.float64):
    """db [N, d], queries [nq, d] -> (scores float32 [nq, k], rows int64 [nq, k])."""
    s = queries.astype(dtype) @ db.astype(dtype).T
    n = db.shape[0]
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]        # stable: ties by ascending row
    scores = np.take_along_axis(s, order, axis=1).astype(np.float32)
    rows = order.astype(np.int64)
    if n < k:
        scores = np.concatenate([scores, np.full((s.shape[0], k - n), FLT_LOWEST, np.float32)], 1)
        rows = np.concatenate([rows, np.full((s.shape[0], k - n), -1, np.int64)], 1)
    return scores, rows


def search_fp32_blas(db: np.ndarray, queries: np.ndarray, k: int):
    """The same through an fp32 sgemm + partial sort — what faiss does on the CPU; used as bench.py's cpu_baseline."""
    s = queries.astype(np.float32) @ db.astype(np.float32).T
    kk = min(k, db.shape[0])
    part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
    ps = np.take_along_axis(s, part, axis=1)
    o = np.lexsort((part, -ps), axis=1)
    return np.take_along_axis(ps, o, axis=1), np.take_along_axis(part, o, axis=1).astype(np.int64)
