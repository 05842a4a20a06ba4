"""MI355X: the flat inner-product index (czc_index_* through the C ABI) against oracle/flat_index_oracle.py.

Integer-valued vectors make every product and partial sum exact in fp32, so scores AND rows must match the oracle bit
for bit, ties included (ascending row).  Real-valued vectors are compared with the float64 oracle within fp32
summation error: |score - exact| <= 2e-6 * sum|q_i d_i| (768 terms, fp32 accumulate), and every returned row must be a
legitimate member of the top k at that tolerance."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import flat_index_oracle as oracle  # noqa: E402


def _index(blocks):
    from czech_contriever_amd.flat_index import FlatIPIndex
    ix = FlatIPIndex(blocks[0].shape[1])
    for b in blocks:
        ix.add(torch.from_numpy(b))
    return ix


@pytest.fixture(params=["coarse", "exact"])
def mode(request):
    """coarse = bf16 scan with error margins + fp32 rescoring (the default); exact = the fp32 matrix-core scan.  Both
    must give the oracle's answer."""
    return request.param


def _search(ix, q, k, mode="coarse"):
    s, r = ix.search(torch.from_numpy(q).cuda(), k, mode=mode)
    torch.cuda.synchronize()
    return s.cpu().numpy(), r.cpu().numpy()


@pytest.mark.parametrize("n_blocks,nq,k,dim", [
    ([70001], 130, 100, 64),            # two chunks (65536 + 4465), ragged query and row tiles
    ([1000, 66000, 300, 140000], 33, 100, 32),   # blocks of uneven size, five chunks
    ([5000], 7, 1, 16),                 # k = 1, smallest dim
    ([3000, 3000], 5, 2048, 48),        # largest k, sort-only path
    ([90000], 3, 2048, 16),             # largest k through the radix select
])
def test_integer_vectors_bit_exact(n_blocks, nq, k, dim, mode):
    rng = np.random.default_rng(sum(n_blocks) + nq + k)
    blocks = [rng.integers(-4, 5, (n, dim)).astype(np.float32) for n in n_blocks]
    q = rng.integers(-4, 5, (nq, dim)).astype(np.float32)
    got_s, got_r = _search(_index(blocks), q, k, mode)
    want_s, want_r = oracle.search(np.concatenate(blocks), q, k)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(got_r, want_r)


@pytest.mark.parametrize("rows", [1, 7, 8, 9, 255, 257, 4099])
def test_tiny_shapes(rows, mode):
    rng = np.random.default_rng(rows)
    for nq, k, dim in ((1, 1, 16), (3, 5, 32), (2, 300, 16)):
        db = rng.integers(-4, 5, (rows, dim)).astype(np.float32)
        q = rng.integers(-4, 5, (nq, dim)).astype(np.float32)
        got_s, got_r = _search(_index([db]), q, k, mode)
        want_s, want_r = oracle.search(db, q, k)
        assert np.array_equal(got_s, want_s) and np.array_equal(got_r, want_r), (rows, nq, k, dim)


def test_duplicated_passages_tie_by_row(mode):
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, (50, 32)).astype(np.float32)
    db = np.tile(base, (1500, 1))                   # 75000 rows, every vector 1500 times: ties far beyond k
    q = rng.integers(-3, 4, (9, 32)).astype(np.float32)
    got_s, got_r = _search(_index([db]), q, 100, mode)
    want_s, want_r = oracle.search(db, q, 100)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(got_r, want_r)


def test_short_index_is_padded_like_faiss(mode):
    rng = np.random.default_rng(1)
    db = rng.integers(-4, 5, (37, 16)).astype(np.float32)
    q = rng.integers(-4, 5, (4, 16)).astype(np.float32)
    got_s, got_r = _search(_index([db]), q, 50, mode)
    want_s, want_r = oracle.search(db, q, 50)
    assert np.array_equal(got_r, want_r) and np.array_equal(got_s, want_s)
    assert (got_r[:, 37:] == -1).all() and (got_s[:, 37:] == oracle.FLT_LOWEST).all()


def test_real_vectors_within_fp32_summation_error(mode):
    rng = np.random.default_rng(11)
    n, dim, nq, k = 150_000, 768, 70, 100
    db = rng.standard_normal((n, dim), dtype=np.float32) * 0.05
    db[1::7] += 0.02                                # uneven score distribution, best rows spread over all chunks
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    got_s, got_r = _search(_index([db[:40_000], db[40_000:]]), q, k, mode)
    exact = q.astype(np.float64) @ db.astype(np.float64).T
    bound = 2e-6 * (np.abs(q).astype(np.float64) @ np.abs(db).astype(np.float64).T)
    tol = np.take_along_axis(bound, got_r, 1)
    assert (got_r >= 0).all() and all(len(set(r)) == k for r in got_r)
    assert (np.abs(got_s - np.take_along_axis(exact, got_r, 1)) <= tol).all()
    assert (np.diff(got_s, axis=1) <= 0).all()
    want_s, want_r = oracle.search(db, q, k)
    kth = want_s[:, -1:].astype(np.float64)
    assert (np.take_along_axis(exact, got_r, 1) >= kth - 2 * bound.max()).all()   # nothing outside the top k was returned
    assert (got_r == want_r).mean() > 0.99          # and apart from near-ties the order is the oracle's


def test_survivor_overflow_falls_back_to_the_fp32_scan():
    """70 000 copies of one vector (+ a few better rows late in the index): every row is within the bf16 margin of the
    k-th best, the survivor lists overflow, the flagged queries are redone by the fp32 scan — still the oracle's answer."""
    rng = np.random.default_rng(9)
    v = rng.integers(-3, 4, (1, 32)).astype(np.float32)
    db = np.repeat(v, 70_000, 0)
    db[[100, 66_000, 69_999]] *= 2.0
    q = np.concatenate([v, -v, rng.integers(-3, 4, (3, 32)).astype(np.float32)])
    ix = _index([db])
    qd = torch.from_numpy(q).cuda()
    _, _, flags = ix._search_coarse(qd, 100)
    assert int(flags.sum()) >= 2                      # q = v and q = -v tie on (nearly) every row
    got_s, got_r = _search(ix, q, 100, "coarse")
    want_s, want_r = oracle.search(db, q, 100)
    assert np.array_equal(got_s, want_s) and np.array_equal(got_r, want_r)


def test_indexer_surface_and_file_round_trip(tmp_path):
    from src.index import Indexer
    rng = np.random.default_rng(3)
    emb = rng.integers(-4, 5, (1200, 32)).astype(np.float32)
    ids = [f"p{i}" for i in range(1200)]
    ix = Indexer(32)
    ix.index_data(ids[:500], emb[:500].astype(np.float16))   # fp16 shards as `model.half()` writes them
    ix.index_data(ids[500:], emb[500:])
    q = rng.integers(-4, 5, (5, 32)).astype(np.float32)
    res = ix.search_knn(q, 10, index_batch_size=2)           # three batches
    want_s, want_r = oracle.search(emb, q, 10)
    assert len(res) == 5
    for (db_ids, scores), ws, wr in zip(res, want_s, want_r):
        assert db_ids == [f"p{i}" for i in wr] and np.array_equal(scores, ws) and scores.dtype == np.float32
    ix.serialize(str(tmp_path))
    iy = Indexer(32)
    iy.deserialize_from(str(tmp_path))
    res2 = iy.search_knn(q, 10)
    assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    with pytest.raises(NotImplementedError):
        Indexer(32, n_subquantizers=8)
