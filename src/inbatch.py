"""`src.inbatch` for the MI355X path: the in-batch-negatives objective on libczc.so.

Surface kept from Ajchler/czech-contriever src/inbatch.py:17-90 — `InBatch(opt, retriever=None, tokenizer=None)`
reading opt.{norm_doc, norm_query, label_smoothing, temperature, retriever_model_id, pooling, random_init};
`get_encoder()`; `forward(q_tokens, q_mask, k_tokens, k_mask, stats_prefix="", iter_stats={}, **kwargs)
-> (loss, iter_stats)`; attributes `tokenizer`, `encoder` (state_dict keys `encoder.*` with HF BERT names).

ONE encoder embeds queries and keys, both with gradients: two differentiable forwards of the same
FlatBertEncoder are pending when backward starts (engine.FlatBertEncoder._lease_activations keeps their
activation workspaces apart) and both backwards accumulate into the same flat gradient array.  The keys of all
ranks are concatenated in rank order by `dist_utils.gather` — an all-gather whose backward all-reduces the
gradient and keeps this rank's slice (reference src/dist_utils.py:7-26) — so a query's positive sits at column
rank * B + i (reference :73).  The [B, W*B] score product, the cross-entropy with label smoothing, the four stats
and the gradients w.r.t. queries and keys are one libczc call (czc_inbatch_head, csrc/head.hip: fp32, fixed summation
order)."""
from __future__ import annotations

import torch
import torch.nn as nn

from czech_contriever_amd.engine import inbatch_loss
from src import contriever, dist_utils
from src.moco import MoCo, resolve_model_config


class InBatch(nn.Module):
    def __init__(self, opt, retriever=None, tokenizer=None):
        super().__init__()
        self.opt = opt
        self.norm_doc = opt.norm_doc
        self.norm_query = opt.norm_query
        self.label_smoothing = opt.label_smoothing
        if retriever is None or tokenizer is None:
            retriever, tokenizer = self._load_retriever(opt.retriever_model_id, pooling=opt.pooling,
                                                        random_init=opt.random_init)
        self.tokenizer = tokenizer
        self.encoder = retriever
        self.encoder.load_dropout_state({"salt": 0})   # the dropout stream belongs to the role, not to construction order

    def _load_retriever(self, model_id, pooling, random_init):
        if "xlm" in model_id:
            raise NotImplementedError("XLMRetriever is outside the MI355X path (bert-family encoders only)")
        cfg, tokenizer = resolve_model_config(model_id)
        retriever = contriever.Contriever(cfg, pooling=pooling)
        if not random_init:
            MoCo._load_local_weights(retriever, model_id)
        retriever.config.pooling = pooling
        return retriever, tokenizer

    def get_encoder(self):
        return self.encoder

    def set_dropout_position(self, micro_steps_done: int):
        """see MoCo.set_dropout_position: the one encoder embeds queries and keys, two training forwards per micro-step"""
        self.encoder.load_dropout_state({"calls": 2 * int(micro_steps_done)})

    def forward(self, q_tokens, q_mask, k_tokens, k_mask, stats_prefix="", iter_stats={}, **kwargs):
        bsz = len(q_tokens)
        qemb = self.encoder(input_ids=q_tokens, attention_mask=q_mask, normalize=self.norm_query)
        kemb = self.encoder(input_ids=k_tokens, attention_mask=k_mask, normalize=self.norm_doc)
        all_keys = dist_utils.gather(kemb)
        # scores = (qemb / T) @ all_keys^T, labels = rank * B + arange(B), F.cross_entropy(label_smoothing) and the four
        # stats (:71-88) in one libczc call; the stats stay 0-d device tensors instead of .item() round trips
        stats = torch.empty(8, dtype=torch.float32, device=qemb.device)
        loss = inbatch_loss(qemb, all_keys, dist_utils.get_rank() * len(kemb), self.opt.temperature, self.label_smoothing, stats)
        prefix = stats_prefix + "/" if len(stats_prefix) > 0 else ""
        iter_stats[f"{prefix}loss"] = (loss.detach(), bsz)
        iter_stats[f"{prefix}accuracy"] = (stats[1], bsz)
        iter_stats[f"{prefix}stdq"] = (stats[2], bsz)
        iter_stats[f"{prefix}stdk"] = (stats[3], bsz)
        return loss, iter_stats
