"""HBM-resident flat inner-product index: host side of czc_index_* (include/czc.h).

The index is a list of fp32 [rows, dim] device blocks (one per `add`, as faiss' IndexFlatIP grows by `add`,
reference src/index.py:25-31); a search walks them in order on the current stream and keeps, per query, the
running k best (czech_contriever_amd/csrc/flat_index.hip).  There is no CPU path."""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import lib as L

F32, I32, I64, BF16 = torch.float32, torch.int32, torch.int64, torch.bfloat16


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class FlatIPIndex:
    def __init__(self, dim: int, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("the flat index lives in MI355X HBM (no CPU path)")
        if dim % 16 != 0:
            raise ValueError(f"vector size {dim}: the gfx950 scan kernel needs a multiple of 16")
        self.dim = dim
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.blocks: List[torch.Tensor] = []
        self.coarse: List[Tuple[torch.Tensor, torch.Tensor]] = []   # per block: bf16 image [ceil8(rows), dim], row norms
        self.ntotal = 0
        self._state = None   # (nq, k) -> candidate lists, reused between batches
        self._cstate = None
        self._table = None   # device arrays of block pointers / first rows for czc_index_rescore

    def add(self, x: torch.Tensor):
        """x: fp32 [n, dim], host or device; copied into its own HBM block."""
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}], got {tuple(x.shape)}")
        if x.shape[0] == 0:
            return
        if self.ntotal + x.shape[0] >= 2 ** 31:
            raise ValueError("flat index holds fewer than 2^31 vectors")
        blk = (x.to(device=self.device, dtype=F32).contiguous().clone() if x.is_cuda
               else x.to(dtype=F32).contiguous().to(self.device))
        self.blocks.append(blk)
        self.coarse.append(self._prepare(blk, 1.0))
        self.ntotal += x.shape[0]
        self._table = None

    def _prepare(self, x: torch.Tensor, scale: float):
        """bf16 image (padded to 8 rows) and scale * row norms of an fp32 [n, dim] device tensor (czc_index_prepare)."""
        n = x.shape[0]
        xb = torch.empty(_pad8(n), self.dim, dtype=BF16, device=self.device)
        norms = torch.empty(_pad8(n), dtype=F32, device=self.device)
        L.check(L.load().czc_index_prepare(L.ptr(x, F32), n, self.dim, scale, L.ptr(xb, BF16), L.ptr(norms, F32), L.stream_ptr()))
        return xb, norms

    def _lists(self, nq: int, k: int):
        lib = L.load()
        cap = lib.czc_index_candidate_capacity(k)
        if cap < 0:
            raise ValueError(f"k={k} outside 1..2048")
        key = (nq, cap)
        if self._state is None or self._state[0] != key:
            self._state = (key, torch.empty(nq, cap, dtype=F32, device=self.device),
                           torch.empty(nq, cap, dtype=I32, device=self.device),
                           torch.empty(nq, dtype=I32, device=self.device), torch.empty(nq, dtype=F32, device=self.device))
        return (cap,) + self._state[1:]

    def search(self, queries: torch.Tensor, k: int, mode: str = "coarse") -> Tuple[torch.Tensor, torch.Tensor]:
        """queries fp32 [nq, dim] on the device -> (scores fp32 [nq,k] descending, rows int64 [nq,k]); both on the device.
        Same result layout as faiss.IndexFlatIP.search (reference src/index.py:43), -1 / -FLT_MAX padded.
        mode "coarse": bf16 scoring with error margins + exact fp32 rescoring of the survivors (same result, several
        times faster; queries whose survivor list overflows are redone by the fp32 scan); "exact": the fp32 scan."""
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}], got {tuple(queries.shape)}")
        if queries.shape[0] > 65535:
            raise ValueError("at most 65535 queries per search call (Indexer.search_knn batches by index_batch_size)")
        q = queries.to(device=self.device, dtype=F32).contiguous()
        if mode == "exact" or not self.blocks:
            return self._search_exact(q, k)
        if mode != "coarse":
            raise ValueError(f"mode {mode!r}")
        scores, rows, flags = self._search_coarse(q, k)
        bad = torch.nonzero(flags).flatten()          # one host sync per search: the results are about to leave the GPU anyway
        if bad.numel():
            s2, r2 = self._search_exact(q[bad].contiguous(), k)
            scores[bad], rows[bad] = s2, r2
        return scores, rows

    def _search_coarse(self, q: torch.Tensor, k: int):
        lib = L.load()
        nq = q.shape[0]
        cap = lib.czc_index_coarse_capacity(k)
        if cap < 0:
            raise ValueError(f"k={k} outside 1..2048")
        key = (nq, cap)
        if self._cstate is None or self._cstate[0] != key:
            dev = self.device
            self._cstate = (key, torch.empty(nq, cap, dtype=F32, device=dev), torch.empty(nq, cap, dtype=F32, device=dev),
                            torch.empty(nq, cap, dtype=I32, device=dev), torch.empty(nq, dtype=I32, device=dev),
                            torch.empty(nq, dtype=F32, device=dev), torch.empty(nq, dtype=I32, device=dev),
                            torch.empty(nq, dtype=F32, device=dev), torch.empty(nq, dtype=I32, device=dev))
        _, cs, ce, ci, cnt, lb, flags, tau, kept = self._cstate
        if self._table is None:
            starts = [0]
            for b in self.blocks:
                starts.append(starts[-1] + b.shape[0])
            self._table = (torch.tensor([b.data_ptr() for b in self.blocks], dtype=I64, device=self.device),
                           torch.tensor(starts, dtype=I64, device=self.device))
        qb, qmargin = self._prepare(q, float(lib.czc_index_coarse_epsilon()))
        st = L.stream_ptr()
        L.check(lib.czc_index_coarse_begin(L.ptr(cnt, I32), L.ptr(lb, F32), L.ptr(flags, I32), L.ptr(kept, I32), nq, st))
        base = 0
        for i, (blk, (xb, norms)) in enumerate(zip(self.blocks, self.coarse)):
            L.check(lib.czc_index_coarse_scan(L.ptr(xb, BF16), L.ptr(norms, F32), blk.shape[0], self.dim, base, L.ptr(qb, BF16),
                                              L.ptr(qmargin, F32), nq, k, int(i == 0), L.ptr(cs, F32), L.ptr(ce, F32),
                                              L.ptr(ci, I32), L.ptr(cnt, I32), L.ptr(lb, F32), L.ptr(flags, I32), L.ptr(kept, I32), cap, st))
            base += blk.shape[0]
        L.check(lib.czc_index_rescore(L.ptr(self._table[0], I64), L.ptr(self._table[1], I64), len(self.blocks), self.dim,
                                      L.ptr(q, F32), nq, k, L.ptr(cs, F32), L.ptr(ci, I32), L.ptr(cnt, I32), L.ptr(tau, F32), cap, st))
        scores = torch.empty(nq, k, dtype=F32, device=self.device)
        rows = torch.empty(nq, k, dtype=I64, device=self.device)
        L.check(lib.czc_index_finish(L.ptr(cs, F32), L.ptr(ci, I32), L.ptr(cnt, I32), cap, nq, k, L.ptr(scores, F32),
                                     L.ptr(rows, I64), st))
        return scores, rows, flags

    def _search_exact(self, q: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        lib = L.load()
        nq = q.shape[0]
        cap, cs, ci, cnt, tau = self._lists(nq, k)
        st = L.stream_ptr()
        L.check(lib.czc_index_begin(L.ptr(cnt, I32), L.ptr(tau, F32), nq, st))
        base = 0
        for i, blk in enumerate(self.blocks):
            L.check(lib.czc_index_scan(L.ptr(blk, F32), blk.shape[0], self.dim, base, L.ptr(q, F32), nq, k, int(i == 0),
                                       L.ptr(cs, F32), L.ptr(ci, I32), L.ptr(cnt, I32), L.ptr(tau, F32), cap, st))
            base += blk.shape[0]
        scores = torch.empty(nq, k, dtype=F32, device=self.device)
        rows = torch.empty(nq, k, dtype=I64, device=self.device)
        L.check(lib.czc_index_finish(L.ptr(cs, F32), L.ptr(ci, I32), L.ptr(cnt, I32), cap, nq, k, L.ptr(scores, F32),
                                     L.ptr(rows, I64), st))
        return scores, rows

    def reconstruct_all(self) -> torch.Tensor:
        return torch.cat(self.blocks, 0) if self.blocks else torch.empty(0, self.dim, dtype=F32, device=self.device)


# ---- `index.faiss` ------------------------------------------------------------------------------------------
# faiss' own IndexFlatIP file layout (faiss/impl/index_write.cpp: fourcc "IxFI", d, ntotal, two dummies, is_trained,
# metric_type 0 = inner product, then the float count and the floats), so that either side can read the other's file.
# faiss is absent from this image: the layout is restated from its published source and only round-trips with itself
# in the tests.
_HEAD = "<iqqqBi"


def write_faiss_flat(index: FlatIPIndex, path: str) -> None:
    import struct
    with open(path, "wb") as f:
        f.write(b"IxFI")
        f.write(struct.pack(_HEAD, index.dim, index.ntotal, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", index.ntotal * index.dim))
        for blk in index.blocks:
            f.write(blk.cpu().numpy().tobytes())


def read_faiss_flat(path: str) -> FlatIPIndex:
    import struct
    import numpy as np
    with open(path, "rb") as f:
        if f.read(4) != b"IxFI":
            raise ValueError(f"{path}: not a flat inner-product index")
        d, n, _, _, _, metric = struct.unpack(_HEAD, f.read(struct.calcsize(_HEAD)))
        (count,) = struct.unpack("<Q", f.read(8))
        if metric != 0 or count != n * d:
            raise ValueError(f"{path}: metric {metric}, {count} floats for {n} x {d}")
        index = FlatIPIndex(d)
        step = max(1, (1 << 28) // d)   # 1 GiB of floats per block
        for lo in range(0, n, step):
            rows = min(step, n - lo)
            index.add(torch.from_numpy(np.frombuffer(f.read(rows * d * 4), dtype=np.float32).reshape(rows, d).copy()))
    return index
