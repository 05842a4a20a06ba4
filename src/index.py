"""`src.index.Indexer` for the MI355X path: the flat inner-product passage index in HBM.

Surface kept from Ajchler/czech-contriever src/index.py:15-72: `Indexer(vector_sz, n_subquantizers=0, n_bits=8)`,
`index_data(ids, embeddings)`, `search_knn(query_vectors, top_docs, index_batch_size=2048)` returning
`[(list of str ids, float32 scores), ...]`, `serialize(dir)` / `deserialize_from(dir)` on `index.faiss` +
`index_meta.faiss`.  The reference delegates storage and search to faiss (`IndexFlatIP`, not installed here and not
needed): vectors live in device blocks and `search` is czc_index_scan (czech_contriever_amd/csrc/flat_index.hip).
The product-quantised variant (`n_subquantizers > 0`, faiss.IndexPQ) is not built and refuses loudly."""
from __future__ import annotations

import os
import pickle
import struct
from typing import List, Tuple

import numpy as np
import torch

from czech_contriever_amd.flat_index import FlatIPIndex


class Indexer(object):

    def __init__(self, vector_sz, n_subquantizers=0, n_bits=8):
        if n_subquantizers > 0:
            raise NotImplementedError("product-quantised index (faiss.IndexPQ) is not built on this path; use the flat index")
        self.index = FlatIPIndex(vector_sz)
        self.index_id_to_db_id = []

    def index_data(self, ids, embeddings):
        self._update_id_mapping(ids)
        embeddings = np.ascontiguousarray(embeddings.astype('float32'))
        self.index.add(torch.from_numpy(embeddings))
        print(f'Total data indexed {len(self.index_id_to_db_id)}')

    def search_knn(self, query_vectors: np.array, top_docs: int, index_batch_size: int = 2048) -> List[Tuple[List[object], List[float]]]:
        query_vectors = np.ascontiguousarray(query_vectors.astype('float32'))
        result = []
        nbatch = (len(query_vectors) - 1) // index_batch_size + 1
        for k in range(nbatch):
            start_idx = k * index_batch_size
            end_idx = min((k + 1) * index_batch_size, len(query_vectors))
            q = torch.from_numpy(query_vectors[start_idx:end_idx]).to(self.index.device)
            scores, indexes = self.index.search(q, top_docs)
            scores, indexes = scores.cpu().numpy(), indexes.cpu().numpy()
            # convert to external ids (a -1 row of a short result indexes from the end, as the reference's list lookup does)
            db_ids = [[str(self.index_id_to_db_id[i]) for i in query_top_idxs] for query_top_idxs in indexes]
            result.extend([(db_ids[i], scores[i]) for i in range(len(db_ids))])
        return result

    # `index.faiss` is written in faiss' own IndexFlatIP layout (faiss/impl/index_write.cpp: fourcc "IxFI", d, ntotal,
    # two dummies, is_trained, metric_type 0, then the float count and the floats) so that either side can read the
    # other's file.  faiss is absent from this image: the layout is restated from its published source, unverified.
    def serialize(self, dir_path):
        index_file = os.path.join(dir_path, 'index.faiss')
        meta_file = os.path.join(dir_path, 'index_meta.faiss')
        print(f'Serializing index to {index_file}, meta data to {meta_file}')
        d, n = self.index.dim, self.index.ntotal
        with open(index_file, 'wb') as f:
            f.write(b'IxFI')
            f.write(struct.pack('<iqqqBi', d, n, 1 << 20, 1 << 20, 1, 0))
            f.write(struct.pack('<Q', n * d))
            for blk in self.index.blocks:
                f.write(blk.cpu().numpy().tobytes())
        with open(meta_file, mode='wb') as f:
            pickle.dump(self.index_id_to_db_id, f)

    def deserialize_from(self, dir_path):
        index_file = os.path.join(dir_path, 'index.faiss')
        meta_file = os.path.join(dir_path, 'index_meta.faiss')
        print(f'Loading index from {index_file}, meta data from {meta_file}')
        with open(index_file, 'rb') as f:
            if f.read(4) != b'IxFI':
                raise ValueError(f'{index_file}: not a flat inner-product index')
            d, n, _, _, _, metric = struct.unpack('<iqqqBi', f.read(4 + 8 * 3 + 1 + 4))
            if metric != 0:
                raise ValueError(f'{index_file}: metric {metric} is not inner product')
            (count,) = struct.unpack('<Q', f.read(8))
            if count != n * d:
                raise ValueError(f'{index_file}: {count} floats for {n} x {d}')
            self.index = FlatIPIndex(d)
            step = max(1, (1 << 28) // d)   # 1 GiB of floats at a time
            for s in range(0, n, step):
                rows = min(step, n - s)
                buf = np.frombuffer(f.read(rows * d * 4), dtype=np.float32).reshape(rows, d)
                self.index.add(torch.from_numpy(buf.copy()))
        print('Loaded index of type %s and size %d', type(self.index), self.index.ntotal)
        with open(meta_file, "rb") as reader:
            self.index_id_to_db_id = pickle.load(reader)
        assert len(
            self.index_id_to_db_id) == self.index.ntotal, 'Deserialized index_id_to_db_id should match faiss index size'

    def _update_id_mapping(self, db_ids: List):
        self.index_id_to_db_id.extend(db_ids)
