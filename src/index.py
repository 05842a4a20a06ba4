"""`src.index.Indexer` on MI355X: the flat inner-product passage index, resident in HBM.

Drop-in for the class of the same name in Ajchler/czech-contriever src/index.py:15-72, which wraps faiss:
  Indexer(vector_sz, n_subquantizers=0, n_bits=8)
  .index_data(ids, embeddings)                      append vectors and their external ids
  .search_knn(query_vectors, top_docs, index_batch_size=2048) -> [(ids as str, float32 scores), ...] per query
  .serialize(dir) / .deserialize_from(dir)           index.faiss + index_meta.faiss
  .index_id_to_db_id                                 row -> external id
faiss is neither installed nor needed here: vectors are fp32 blocks on the GPU (czech_contriever_amd.flat_index)
and the search is czc_index_scan, libczc.so's fp32 matrix-core scan with an exact running top-k
(czech_contriever_amd/csrc/flat_index.hip).  Only the flat index is built: `n_subquantizers > 0`
(faiss.IndexPQ, product quantisation) refuses loudly."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from czech_contriever_amd import flat_index


def _f32(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32))


class Indexer(object):
    def __init__(self, vector_sz, n_subquantizers=0, n_bits=8):
        if n_subquantizers > 0:
            raise NotImplementedError("product-quantised index (faiss.IndexPQ) is not built on this path; use the flat index")
        self.index = flat_index.FlatIPIndex(vector_sz)
        self.index_id_to_db_id = []
        self._names = None     # str(id) per row as one object array, built on first search after a change

    def index_data(self, ids, embeddings):
        self.index_id_to_db_id.extend(ids)
        self._names = None
        self.index.add(_f32(embeddings))
        print(f'Total data indexed {len(self.index_id_to_db_id)}')

    def search_knn(self, query_vectors, top_docs: int, index_batch_size: int = 2048):
        queries = _f32(query_vectors)
        if self._names is None or len(self._names) != len(self.index_id_to_db_id):
            # the search itself takes ~40 ms per 2048 queries over 10 M passages: 200 000 str() calls per batch would
            # cost more than that, so the ids are stringified once
            self._names = np.array([str(x) for x in self.index_id_to_db_id], dtype=object)
        result = []
        for lo in range(0, len(queries), index_batch_size):
            part = queries[lo:lo + index_batch_size].to(self.index.device)
            scores, rows = self.index.search(part, top_docs)
            scores, rows = scores.cpu().numpy(), rows.cpu().numpy()
            # a short result's -1 rows index the id list from its end, as the reference's plain list lookup does
            found = self._names[rows].tolist() if len(self._names) else [[] for _ in rows]
            result += list(zip(found, scores))
        return result

    @staticmethod
    def _files(dir_path):
        return os.path.join(dir_path, 'index.faiss'), os.path.join(dir_path, 'index_meta.faiss')

    def serialize(self, dir_path):
        index_file, meta_file = self._files(dir_path)
        print(f'Serializing index to {index_file}, meta data to {meta_file}')
        flat_index.write_faiss_flat(self.index, index_file)
        with open(meta_file, 'wb') as f:
            pickle.dump(self.index_id_to_db_id, f)

    def deserialize_from(self, dir_path):
        index_file, meta_file = self._files(dir_path)
        print(f'Loading index from {index_file}, meta data from {meta_file}')
        self.index = flat_index.read_faiss_flat(index_file)
        with open(meta_file, 'rb') as f:   # the id list this class wrote next to the index
            self.index_id_to_db_id = pickle.load(f)
        self._names = None
        if len(self.index_id_to_db_id) != self.index.ntotal:
            raise ValueError('Deserialized index_id_to_db_id should match faiss index size')
        print(f'Loaded flat index of size {self.index.ntotal}')
