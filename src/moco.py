"""`src.moco` for the MI355X path: the MoCo contrastive module on libczc.so.

Surface kept from Ajchler/czech-contriever src/moco.py:187-356 — `MoCo(opt)` reading
opt.{queue_size, momentum, temperature, label_smoothing, norm_doc, norm_query,
moco_train_mode_encoder_k, weight_decay, weight_decay_from_init, retriever_model_id, pooling,
random_init, projection_size}; `forward(q_tokens, q_mask, k_tokens, k_mask, stats_prefix="",
iter_stats={}, **kwargs) -> (loss, iter_stats)`; `get_encoder(return_encoder_k=False)`;
`init_weights_to_gpu()`; attributes tokenizer / encoder_q / encoder_k; buffers `queue` fp32 [C, K]
and `queue_ptr` int64 [1]; state_dict keys `queue`, `queue_ptr`, `encoder_q.*`, `encoder_k.*`.

Order inside forward is the reference's (:296-353): query encoder (grad) -> momentum EMA -> key
encoder (no grad) -> logits against the queue as it was BEFORE this step -> loss -> all-gather keys
in rank order -> enqueue -> stats.  What differs underneath: the EMA is one launch over the flat
parameter array, the logits/CE never materialise [B, K+1], the queue has a slot-major bf16 working
copy, the pointer never leaves the device, and no stat forces a host sync (values are 0-d device
tensors; `.item()` them when you print)."""
from __future__ import annotations

import copy
import json
import os

import torch
import torch.nn as nn

from czech_contriever_amd import lib as L
from czech_contriever_amd.engine import infonce_loss
from src import contriever, dist_utils

F32, BF16 = torch.float32, torch.bfloat16

_BUILTIN_SHAPES = {
    # offline defaults for ids the reference would fetch: architecture only, weights are random_init
    "bert-base-uncased": dict(vocab_size=30522),
    "bert-base-multilingual-cased": dict(vocab_size=119547),
}


def resolve_model_config(model_id):
    """(config, tokenizer) for `--retriever_model_id`, from LOCAL files only.
    "czert" -> models/czert/config.json + vocab (cwd-relative, as the reference: src/moco.py:227-229)."""
    tokenizer = None
    if "czert" in model_id:
        cfg_dir = "models/czert"
    elif os.path.isdir(model_id):
        cfg_dir = model_id
    else:
        cfg_dir = None
    if cfg_dir is not None:
        path = os.path.join(cfg_dir, "config.json")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (needed for --retriever_model_id {model_id})")
        with open(path) as f:
            cfg = contriever.bert_config(**json.load(f))
        if os.path.exists(os.path.join(cfg_dir, "vocab.txt")):
            try:
                import transformers
                tokenizer = transformers.BertTokenizer.from_pretrained(cfg_dir + "/")
            except Exception:  # tokenizers are only needed by the data pipeline
                tokenizer = None
    elif model_id in _BUILTIN_SHAPES:
        cfg = contriever.bert_config(**_BUILTIN_SHAPES[model_id])
    else:
        raise FileNotFoundError(f"--retriever_model_id {model_id}: no local directory and no built-in shape; "
                                f"models cannot be downloaded on this target")
    if tokenizer is not None and (("bert-" in model_id) or ("czert" in model_id) or ("BERT" in model_id)):
        if tokenizer.bos_token_id is None:
            tokenizer.bos_token = "[CLS]"
        if tokenizer.eos_token_id is None:
            tokenizer.eos_token = "[SEP]"
    return cfg, tokenizer


class _L2FromInit(torch.autograd.Function):
    """weight_decay * sum((p - p_init)^2) over the query encoder (src/moco.py:332-336) and its gradient, each ONE pass of
    libczc over the flat arrays (czc_l2_from_init_forward / _backward, csrc/optim.hip)."""

    @staticmethod
    def forward(ctx, anchor, enc, init_flat, wd):
        ctx.enc, ctx.init_flat, ctx.wd = enc, init_flat, wd
        p = enc.flat_params
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        partials = torch.empty(4096, dtype=torch.float32, device=p.device)
        L.check(L.load().czc_l2_from_init_forward(p.data_ptr(), init_flat.data_ptr(), p.numel(), float(wd), partials.data_ptr(),
                                                  out.data_ptr(), L.stream_ptr()))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        enc = ctx.enc
        for hook in enc._pre_grad_access:   # a gradient all-reduce may still be averaging the array (DataParallel)
            hook()
        fresh = any(p.grad is None for p in enc._plist if p.requires_grad)
        grads = enc.flat_grads()
        if fresh:
            from czech_contriever_amd.engine import _numel
            for (name, shape, off), p in zip(enc._table, enc._plist):
                if p.requires_grad:
                    p.grad = grads[off:off + _numel(shape)].view(shape)
        p = enc.flat_params
        up = g.detach().reshape(1).to(torch.float32).contiguous()
        L.check(L.load().czc_l2_from_init_backward(p.data_ptr(), ctx.init_flat.data_ptr(), p.numel(), float(ctx.wd), up.data_ptr(),
                                                   grads.data_ptr(), 0 if fresh else 1, L.stream_ptr()))
        return torch.zeros_like(g).reshape(1), None, None, None


class MoCo(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.queue_size = opt.queue_size
        self.momentum = opt.momentum
        self.temperature = opt.temperature
        self.label_smoothing = opt.label_smoothing
        self.norm_doc = opt.norm_doc
        self.norm_query = opt.norm_query
        self.moco_train_mode_encoder_k = opt.moco_train_mode_encoder_k
        self.weight_decay = opt.weight_decay
        self.weight_decay_from_init = opt.weight_decay_from_init
        self.key_gather_split = True   # data parallel: last key-encoder layer in two halves, first half's all-gather under it
        # one GPU: the key encoder's pass (EMA + forward, nothing differentiates through it) is independent of the query
        # encoder's forward; on a second HIP stream the two passes' kernels fill each other's launch gaps and tails
        # (interleaved A/B at the headline shapes: 55.79 -> 55.63 ms per step.  Off unless CZC_KEY_OVERLAP=1: 0.3 % is inside
        #  the spread between devices, and kernels of two streams that share the CUs stretch each other's durations, which
        #  makes every per-kernel figure -- hipEvent probes, rocprofv3 -- read high)
        self.key_overlap = os.environ.get("CZC_KEY_OVERLAP", "0") == "1"
        self._key_stream = None
        # The momentum EMA of the NEXT forward applied inside the optimizer's pass (FusedAdamW.attach_ema): the training
        # loop switches it on for steps after which only the next training forward reads the key encoder (a checkpoint
        # written in between would hold key weights one EMA ahead of the reference's; utils.save refuses that).
        self.ema_fold = False
        self._ema_prefolded = None

        retriever, tokenizer = self._load_retriever(opt.retriever_model_id, pooling=opt.pooling,
                                                    random_init=opt.random_init)
        self.tokenizer = tokenizer
        self.encoder_q = retriever
        self.encoder_k = copy.deepcopy(retriever)
        for param_k in self.encoder_k.parameters():
            param_k.requires_grad = False
        # dropout streams pinned to the role, not to how many encoders the process built before (a teacher, a temporary
        # Contriever): query encoder 0, key encoder 1; a resumed run is put at the checkpointed step's position by
        # set_dropout_position (train.py), the checkpoint keeps the reference's five keys
        self.encoder_q.load_dropout_state({"salt": 0})
        self.encoder_k.load_dropout_state({"salt": 1})

        self.register_buffer("queue", nn.functional.normalize(torch.randn(opt.projection_size, self.queue_size), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))

        if opt.weight_decay_from_init:
            self.init_weights = copy.deepcopy(self.encoder_q.state_dict())
            self._init_flat = None

        self._queue_bf16 = None
        self._queue_version = -1
        self._head_ws = None
        self._stats = None
        self._key_exchange = {}
        self._l2_anchor = None

    def _load_retriever(self, model_id, pooling, random_init):
        cfg, tokenizer = resolve_model_config(model_id)
        if "xlm" in model_id:
            raise NotImplementedError("XLMRetriever is outside the MI355X path (bert-family encoders only)")
        retriever = contriever.Contriever(cfg, pooling=pooling)
        if not random_init:
            self._load_local_weights(retriever, model_id)
        retriever.config.pooling = pooling
        return retriever, tokenizer

    @staticmethod
    def _load_local_weights(retriever, model_id):
        folder = "models/czert" if "czert" in model_id else model_id
        for fname in ("model.safetensors", "pytorch_model.bin"):
            path = os.path.join(folder, fname)
            if os.path.exists(path):
                if fname.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    state = load_file(path)
                else:
                    state = torch.load(path, map_location="cpu", weights_only=True)
                state = {k[len("bert."):] if k.startswith("bert.") else k: v for k, v in state.items()}
                own = retriever.state_dict()
                retriever.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)
                return
        raise FileNotFoundError(f"no local weights for {model_id} (use --random_init; downloads are impossible here)")

    def get_encoder(self, return_encoder_k=False):
        return self.encoder_k if return_encoder_k else self.encoder_q

    def set_dropout_position(self, micro_steps_done: int):
        """A resumed run continues the dropout streams where the checkpointed run stood instead of replaying the masks of
        steps 0..n: each encoder draws one mask set per training forward, i.e. per micro-step (train.py calls this with
        step * update_freq after utils.load; the checkpoint keeps the reference's five keys)."""
        self.encoder_q.load_dropout_state({"calls": int(micro_steps_done)})
        self.encoder_k.load_dropout_state({"calls": int(micro_steps_done)})

    # ------------------------------------------------------------------------------------------
    def _queue_mirror(self):
        """slot-major bf16 working copy [K, C] of `queue` [C, K]; rebuilt when torch wrote the buffer."""
        q = self.queue
        if not q.is_cuda:
            raise RuntimeError("libczc has no CPU path: call model.cuda() first")
        if self._queue_bf16 is None or self._queue_bf16.device != q.device or self._queue_version != q._version:
            C_, K = q.shape
            self._queue_bf16 = torch.empty(K, C_, dtype=BF16, device=q.device)
            L.check(L.load().czc_queue_import(q.contiguous().data_ptr(), C_, K, self._queue_bf16.data_ptr(), L.stream_ptr()))
            self._queue_version = q._version
        return self._queue_bf16

    def allow_ema_fold(self, ok: bool):
        """train.py / distill.py / bench.py, before optimizer.step(): True when nothing but the next training forward will
        look at the key encoder after this step (no checkpoint, not the last step)."""
        self.ema_fold = bool(ok)

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """param_k = param_k * m + param_q * (1 - m) for every parameter: one launch (reference :260-269)."""
        ek, eq = self.encoder_k, self.encoder_q
        if self._ema_prefolded is not None:
            stamp, self._ema_prefolded = self._ema_prefolded, None
            if stamp == (eq._write_stamp(), ek._write_stamp()):
                return      # the optimizer's pass already applied THIS forward's EMA (czc_adamw_ema_step), bit for bit
        mirror = ek.mirror()
        L.check(L.load().czc_ema_update(ek.flat_params.data_ptr(), eq.flat_params.data_ptr(), ek.flat_params.numel(),
                                        float(self.momentum), float(1.0 - self.momentum), mirror.data_ptr(),
                                        mirror.numel(), L.stream_ptr()))
        ek.mark_mirror_fresh()

    @torch.no_grad()
    def _enqueue_rows(self, keys):
        """overwrite queue[:, ptr:ptr+n] with keys [n, C] (all ranks, rank order) and advance the pointer, on the device"""
        batch_size = keys.shape[0]
        assert self.queue_size % batch_size == 0, f"{batch_size}, {self.queue_size}"  # for simplicity
        qb = self._queue_mirror()
        L.check(L.load().czc_enqueue(keys.contiguous().data_ptr(), batch_size, keys.shape[1], self.queue_size,
                                     qb.data_ptr(), self.queue.data_ptr(), self.queue_ptr.data_ptr(), L.stream_ptr()))

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, process_group=None):
        """all-gather keys (rank order, over `process_group`) and overwrite queue[:, ptr:ptr+n] (reference :272-287).
        Stand-alone form: `_step` splits it into the exchange started right behind the key encoder and the enqueue
        behind the logits (`_exchange`)."""
        if dist_utils.get_world_size(process_group) > 1:
            kx = self._exchange(process_group)
            kx.gather(keys.contiguous(), 0)
            kx.enqueue(self._enqueue_rows)
        else:
            self._enqueue_rows(keys)

    def _exchange(self, process_group):
        from czech_contriever_amd.parallel import KeyExchange
        kx = self._key_exchange.get(process_group)
        if kx is None:
            kx = self._key_exchange[process_group] = KeyExchange(process_group)
        return kx

    def _compute_logits(self, q, k):
        """Materialised logits [B, 1+K] (debug / evaluation only; the training path never builds them)."""
        l_pos = torch.einsum("nc,nc->n", [q, k]).unsqueeze(-1)
        l_neg = torch.einsum("nc,ck->nk", [q, self.queue.clone().detach()])
        return torch.cat([l_pos, l_neg], dim=1)

    def forward(self, q_tokens, q_mask, k_tokens, k_mask, stats_prefix="", iter_stats={}, **kwargs):
        loss, _, iter_stats = self._step(q_tokens, q_mask, k_tokens, k_mask, None, stats_prefix, iter_stats)
        return loss, iter_stats

    def _step(self, q_tokens, q_mask, k_tokens, k_mask, process_group, stats_prefix, iter_stats):
        bsz = q_tokens.size(0)
        w = dist_utils.get_world_size(process_group)
        kx = self._exchange(process_group) if w > 1 else None
        side = None
        if self.key_overlap and kx is None and q_tokens.is_cuda:
            main = torch.cuda.current_stream()
            if self._key_stream is None or self._key_stream.device != q_tokens.device:
                self._key_stream = torch.cuda.Stream(device=q_tokens.device)
            side = self._key_stream
            side.wait_stream(main)      # the optimizer's update of the query weights, the previous enqueue
            with torch.cuda.stream(side), torch.no_grad():
                self._momentum_update_key_encoder()
                if not self.encoder_k.training and not self.moco_train_mode_encoder_k:
                    self.encoder_k.eval()
                k = self.encoder_k(input_ids=k_tokens, attention_mask=k_mask, normalize=self.norm_doc)
        q = self.encoder_q(input_ids=q_tokens, attention_mask=q_mask, normalize=self.norm_query)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            k.record_stream(torch.cuda.current_stream())

        with torch.no_grad():
            if side is None:
                self._momentum_update_key_encoder()
                if not self.encoder_k.training and not self.moco_train_mode_encoder_k:
                    self.encoder_k.eval()
            if side is not None:
                pass
            elif kx is not None:
                # cross-GPU negatives: the first half of the keys is on its way to the other ranks while the key encoder
                # finishes the second half; the second half travels under the InfoNCE head below (KeyExchange)
                n_first = kx.split_for(bsz, k_tokens.size(1)) if self.key_gather_split else 0
                kx._events()
                k = self.encoder_k.encode(k_tokens, k_mask, normalize=self.norm_doc, split=(n_first, kx.split_event))
                kx.gather(k, n_first)
            else:
                k = self.encoder_k(input_ids=k_tokens, attention_mask=k_mask, normalize=self.norm_doc)

        qb = self._queue_mirror()
        if self._head_ws is None or self._head_ws_key != (bsz, q.shape[1], self.queue_size, q.device):
            n = L.load().czc_infonce_workspace_bytes(bsz, q.shape[1], self.queue_size)
            self._head_ws = torch.empty(n, dtype=torch.uint8, device=q.device)
            self._head_ws_key = (bsz, q.shape[1], self.queue_size, q.device)
        stats = torch.empty(8, dtype=F32, device=q.device)
        loss = infonce_loss(q, k, qb, self.temperature, self.label_smoothing, self._head_ws, stats)

        if self.weight_decay_from_init:
            if self._init_flat is None or self._init_flat.device != q.device:
                self._init_flat = torch.empty_like(self.encoder_q.flat_params)
                tmp = contriever.Contriever(self.encoder_q.config)
                tmp.load_state_dict(self.init_weights)
                self._init_flat.copy_(tmp.flat_params)
            if self._l2_anchor is None or self._l2_anchor.device != q.device:
                self._l2_anchor = torch.zeros(1, device=q.device, requires_grad=True)
            loss = loss + _L2FromInit.apply(self._l2_anchor, self.encoder_q, self._init_flat, float(self.weight_decay))

        # the enqueue overwrites queue columns this step's logits have just read (they must not contain this step's keys,
        # reference :323 -> :338): it is ordered behind the head, and the next reader of the queue behind it
        if kx is not None:
            kx.enqueue(self._enqueue_rows)
        else:
            self._enqueue_rows(k)

        if len(stats_prefix) > 0:
            stats_prefix = stats_prefix + "/"
        iter_stats[f"{stats_prefix}loss"] = (loss.detach(), bsz)
        iter_stats[f"{stats_prefix}accuracy"] = (stats[1], bsz)
        iter_stats[f"{stats_prefix}stdq"] = (stats[2], bsz)
        iter_stats[f"{stats_prefix}stdk"] = (stats[3], bsz)
        return loss, q, iter_stats

    def init_weights_to_gpu(self):
        self.init_weights = {k: v.cuda() for k, v in self.init_weights.items()}
        self._init_flat = None


class MoCoDistill(MoCo):
    """The student of the teacher/student run (reference src/moco.py:15-184, driven by distill.py): MoCo whose key
    all-gather runs over the students' process group and whose forward also hands back the query embeddings, still
    attached to the graph, for the similarity-matrix loss against the teacher's (distill.py:263-276)."""

    def forward(self, q_tokens, q_mask, k_tokens, k_mask, process_group, stats_prefix="", iter_stats={}, **kwargs):
        return self._step(q_tokens, q_mask, k_tokens, k_mask, process_group, stats_prefix, iter_stats)
