"""`src.contriever` for the MI355X path: the Contriever encoder (BERT + pooling) on libczc.so.

Surface kept from the reference (src/contriever.py:11-55, 104-166 in Ajchler/czech-contriever):
`Contriever(config, pooling="average")`, `forward(input_ids=, attention_mask=, ..., normalize=False)`
returning the pooled embedding [B, H], `config.pooling`, HF BertModel parameter names in
`state_dict()`, and `load_retriever` / `save_contriever` for `checkpoint.pth` files whose "model"
entry prefixes the query encoder with `encoder_q.` (reference :126-135)."""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from czech_contriever_amd.engine import FlatBertEncoder


class Contriever(FlatBertEncoder):
    def __init__(self, config, pooling="average", **kwargs):
        super().__init__(config, pooling=pooling, _init=kwargs.pop("_init", True))
        self._emb_dtype = torch.float32

    def half(self):
        """The reference's inference scripts call `model.half()` (generate_passage_embeddings.py:74-76,
        passage_retrieval.py:153-155).  Here the matrix products already run in bf16 with fp32 accumulation from fp32
        master weights, so nothing is cast: only the returned embeddings become fp16, as the reference's are."""
        self._emb_dtype = torch.float16
        return self

    def float(self):
        self._emb_dtype = torch.float32
        return self

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, output_attentions=None,
                output_hidden_states=None, normalize=False):
        if token_type_ids is not None:
            # tokenizers hand back all-zero segment ids for single sentences; row 0 of the table is what the path adds
            if bool(token_type_ids.any()):
                raise NotImplementedError("the MI355X encoder path adds token type 0 only (got non-zero token_type_ids)")
            token_type_ids = None
        unsupported = dict(token_type_ids=token_type_ids, position_ids=position_ids, head_mask=head_mask,
                           inputs_embeds=inputs_embeds, encoder_hidden_states=encoder_hidden_states,
                           encoder_attention_mask=encoder_attention_mask)
        bad = [k for k, v in unsupported.items() if v is not None]
        if bad or output_attentions or output_hidden_states:
            raise NotImplementedError(f"the MI355X encoder path takes input_ids + attention_mask only (got {bad})")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        emb = self.encode(input_ids, attention_mask, normalize=normalize)
        return emb if self._emb_dtype == torch.float32 else emb.to(self._emb_dtype)


def _config_from(obj):
    """Accept a transformers BertConfig, a dict, or a path to config.json."""
    if isinstance(obj, (str, os.PathLike)):
        import json
        with open(obj) as f:
            obj = json.load(f)
    if isinstance(obj, dict):
        d = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, hidden_act="gelu", initializer_range=0.02)
        d.update(obj)
        return SimpleNamespace(**d)
    return obj


def bert_config(**kw):
    """A minimal stand-in for transformers.BertConfig (same attribute names, bert-base defaults)."""
    return _config_from(dict(kw))


def load_retriever(model_path, pooling="average", random_init=False):
    """Loads `<model_path>/checkpoint.pth` as written by src.utils.save (reference src/contriever.py:104-144)."""
    path = os.path.join(model_path, "checkpoint.pth")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: only local czc/contriever checkpoints can be loaded offline")
    from src.utils import read_checkpoint
    ckpt = read_checkpoint(path)
    opt = ckpt["opt"]
    state = ckpt["model"]
    if any("encoder_q." in key for key in state):
        state = {k.replace("encoder_q.", ""): v for k, v in state.items() if "encoder_q." in k}
    from src.moco import resolve_model_config
    cfg, tokenizer = resolve_model_config(getattr(opt, "retriever_model_id", "czert"))
    retriever = Contriever(cfg, pooling=getattr(opt, "pooling", pooling))
    retriever.load_state_dict(state, strict=False)
    return retriever, tokenizer, getattr(opt, "retriever_model_id", None)
