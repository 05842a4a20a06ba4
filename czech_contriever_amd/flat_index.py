"""HBM-resident flat inner-product index: host side of czc_index_* (include/czc.h).

The index is a list of fp32 [rows, dim] device blocks (one per `add`, as faiss' IndexFlatIP grows by `add`,
reference src/index.py:25-31); a search walks them in order on the current stream and keeps, per query, the
running k best (czech_contriever_amd/csrc/flat_index.hip).  There is no CPU path."""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import lib as L

F32, I32, I64 = torch.float32, torch.int32, torch.int64


class FlatIPIndex:
    def __init__(self, dim: int, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("the flat index lives in MI355X HBM (no CPU path)")
        if dim % 16 != 0:
            raise ValueError(f"vector size {dim}: the gfx950 scan kernel needs a multiple of 16")
        self.dim = dim
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.blocks: List[torch.Tensor] = []
        self.ntotal = 0
        self._state = None   # (nq, k) -> candidate lists, reused between batches

    def add(self, x: torch.Tensor):
        """x: fp32 [n, dim], host or device; copied into its own HBM block."""
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}], got {tuple(x.shape)}")
        if x.shape[0] == 0:
            return
        if self.ntotal + x.shape[0] >= 2 ** 31:
            raise ValueError("flat index holds fewer than 2^31 vectors")
        self.blocks.append(x.to(device=self.device, dtype=F32).contiguous().clone() if x.is_cuda
                           else x.to(dtype=F32).contiguous().to(self.device))
        self.ntotal += x.shape[0]

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

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries fp32 [nq, dim] on the device -> (scores fp32 [nq,k] descending, rows int64 [nq,k]); both on the device.
        Same result layout as faiss.IndexFlatIP.search (reference src/index.py:43), -1 / -FLT_MAX padded."""
        lib = L.load()
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}], got {tuple(queries.shape)}")
        nq = queries.shape[0]
        q = queries.to(device=self.device, dtype=F32).contiguous()
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
