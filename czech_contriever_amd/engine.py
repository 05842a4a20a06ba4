"""Host-side engine: flat parameter storage with HF BERT names on top, autograd glue, workspaces.

The encoder keeps ALL parameters in one fp32 device array (layout: include/czc.h) so that the
momentum EMA, AdamW, the bf16 mirror refresh and the gradient all-reduce are each ONE pass over
contiguous memory.  `state_dict()` still shows the HF BertModel names/shapes the reference's
checkpoints use: every named parameter is a VIEW into the flat array.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import lib as L

BF16, F32 = torch.bfloat16, torch.float32


# ----------------------------------------------------------------------------------------------
# HF name <-> flat offset
# ----------------------------------------------------------------------------------------------
def hf_param_table(shape: L.BertShape) -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(hf_name, shape, flat_offset)] in HF BertModel(add_pooling_layer=False).parameters() order."""
    lib = L.load()
    H, I = shape.hidden_size, shape.intermediate_size
    sp = C.byref(shape)
    off = lambda kind, l=0: int(lib.czc_param_offset(sp, L.P[kind], l))
    t = [
        ("embeddings.word_embeddings.weight", (shape.vocab_size, H), off("WORD")),
        ("embeddings.position_embeddings.weight", (shape.max_position, H), off("POS")),
        ("embeddings.token_type_embeddings.weight", (shape.type_vocab_size, H), off("TYPE")),
        ("embeddings.LayerNorm.weight", (H,), off("ELNW")),
        ("embeddings.LayerNorm.bias", (H,), off("ELNB")),
    ]
    for l in range(shape.num_layers):
        p = f"encoder.layer.{l}."
        wqkv, bqkv = off("WQKV", l), off("BQKV", l)
        t += [
            (p + "attention.self.query.weight", (H, H), wqkv),
            (p + "attention.self.query.bias", (H,), bqkv),
            (p + "attention.self.key.weight", (H, H), wqkv + H * H),
            (p + "attention.self.key.bias", (H,), bqkv + H),
            (p + "attention.self.value.weight", (H, H), wqkv + 2 * H * H),
            (p + "attention.self.value.bias", (H,), bqkv + 2 * H),
            (p + "attention.output.dense.weight", (H, H), off("WO", l)),
            (p + "attention.output.dense.bias", (H,), off("BO", l)),
            (p + "attention.output.LayerNorm.weight", (H,), off("LN1W", l)),
            (p + "attention.output.LayerNorm.bias", (H,), off("LN1B", l)),
            (p + "intermediate.dense.weight", (I, H), off("W1", l)),
            (p + "intermediate.dense.bias", (I,), off("B1", l)),
            (p + "output.dense.weight", (H, I), off("W2", l)),
            (p + "output.dense.bias", (H,), off("B2", l)),
            (p + "output.LayerNorm.weight", (H,), off("LN2W", l)),
            (p + "output.LayerNorm.bias", (H,), off("LN2B", l)),
        ]
    return t


class _Node(nn.Module):
    """A bare container so that parameters can sit under HF's dotted module paths."""


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class FlatBertEncoder(nn.Module):
    """BERT encoder + pooling whose arithmetic runs in libczc.so.  Subclassed by src.contriever.Contriever."""

    _instances = 0          # default dropout salt of an encoder nobody has given one: construction order in the process

    def __init__(self, config, pooling: str = "average", _init: bool = True):
        """_init=False (used by __deepcopy__): the weights are about to be overwritten, draw nothing from the torch RNG."""
        super().__init__()
        self.config = config
        if not hasattr(config, "pooling"):
            self.config.pooling = pooling
        self._shape = L.shape_struct(config)
        lib = L.load()
        sp = C.byref(self._shape)
        self._n = int(lib.czc_param_count(sp))
        self._n_mat = int(lib.czc_matrix_count(sp))
        if self._n <= 0:
            raise RuntimeError("libczc: bad encoder shape")
        if config.hidden_size % 64 or config.hidden_size != config.num_attention_heads * 64:
            raise RuntimeError("libczc builds head_dim 64 only (hidden_size = 64 * num_attention_heads)")
        if getattr(config, "hidden_act", "gelu") != "gelu":
            raise RuntimeError("libczc builds the erf-GELU ('gelu') activation only")
        self._table = hf_param_table(self._shape)
        self._flat = torch.zeros(self._n, dtype=F32)
        self._flat_grad = None
        self._mirror = None
        self._mirror_version = -1
        self._mirror_epoch = 0
        self._mirror_t = None
        self._mirror_t_epoch = -1
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._ws_pool: Dict[tuple, list] = {}
        self._scratch = None
        self._call = 0                            # forwards so far (activation-lease stamp)
        self._drop_calls = 0                      # TRAINING forwards so far: with `_salt`, the dropout stream's position.
        self._salt = FlatBertEncoder._instances   # Owners that checkpoint set both (dropout_state / load_dropout_state):
        FlatBertEncoder._instances += 1           # MoCo pins encoder_q to salt 0 and encoder_k to 1 whatever was built before
        self._anchor = None
        self._pre_grad_access = []  # callables() run before the main stream touches the flat gradient array (comm wait)
        self._post_backward = []  # callables(enc) run right after the backward has been enqueued
        self._range_hooks = []    # callables(enc, elem_lo, elem_hi): a gradient range [lo, hi) of the flat array is final
        self.backward_chunk_layers = 0   # > 0: issue the backward in ranges of this many layers (for comm overlap)
        self._plist: List[nn.Parameter] = []
        for name, shape, off in self._table:
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            prm = nn.Parameter(self._flat[off:off + _numel(shape)].view(shape))
            node.register_parameter(parts[-1], prm)
            self._plist.append(prm)
        if _init:
            self.reset_parameters()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_mirror())

    # ---- initialisation: HF BertPreTrainedModel._init_weights (N(0, initializer_range); LN = 1, 0) ----
    def reset_parameters(self):
        std = float(getattr(self.config, "initializer_range", 0.02))
        with torch.no_grad():
            for (name, shape, off), p in zip(self._table, self._plist):
                if "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(mean=0.0, std=std)
        self.invalidate_mirror()

    # ---- storage management ----------------------------------------------------------------------
    def _rebind(self):
        for (name, shape, off), p in zip(self._table, self._plist):
            p.data = self._flat[off:off + _numel(shape)].view(shape)
            if p.grad is not None:
                p.grad = None if self._flat_grad is None else self._flat_grad[off:off + _numel(shape)].view(shape)

    def _apply(self, fn, recurse=True):
        new = fn(self._flat)
        if new.dtype != F32:
            raise RuntimeError("the MI355X encoder keeps fp32 master weights (bf16 is the compute mirror): dtype casts are refused")
        if new is not self._flat:
            moved = new.device != self._flat.device
            self._flat = new.contiguous()
            if moved:
                self._flat_grad = None
                self._mirror = None
                self._mirror_t = None
                self._ws.clear()
                self._ws_pool.clear()
                self._scratch = None
                self._anchor = None
            self._rebind()
            self.invalidate_mirror()
        return self

    def __deepcopy__(self, memo):
        import copy
        # no random init of the copy: it would consume the torch RNG the reference draws the initial queue from right
        # after copy.deepcopy (src/moco.py:209-219)
        new = self.__class__(copy.deepcopy(self.config, memo), _init=False)
        new.to(self._flat.device)
        with torch.no_grad():
            new._flat.copy_(self._flat)
        for p_new, p_old in zip(new._plist, self._plist):
            p_new.requires_grad = p_old.requires_grad
        new.train(self.training)
        memo[id(self)] = new
        return new

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    @property
    def device(self):
        return self._flat.device

    def flat_grads(self, create=True):
        if self._flat_grad is None and create:
            self._flat_grad = torch.zeros_like(self._flat)
        return self._flat_grad

    def _write_stamp(self) -> int:
        """Changes whenever torch wrote the weights in place: through the flat array or through any parameter view.  (After
        .cuda() / .to() the parameters are re-pointed with `p.data = view`, which gives each its OWN version counter, so the
        flat array's counter alone would miss load_state_dict / p.copy_() / a torch optimizer on a GPU-resident encoder.)"""
        return self._flat._version + sum(p._version for p in self._plist)

    def invalidate_mirror(self):
        """Call after writing the fp32 weights by any means this class cannot see (raw pointers, custom kernels)."""
        self._mirror_version = -1

    def mirror(self) -> torch.Tensor:
        """bf16 copy of the matrix region, refreshed when the fp32 array was written by torch."""
        if not self._flat.is_cuda:
            raise RuntimeError("libczc has no CPU path: move the model to the GPU (model.cuda()) first")
        if self._mirror is None:
            self._mirror = torch.empty(self._n_mat, dtype=BF16, device=self._flat.device)
            self._mirror_version = -1
        stamp = self._write_stamp()
        if self._mirror_version != stamp:
            L.check(L.load().czc_cast_bf16(self._flat.data_ptr(), self._mirror.data_ptr(), self._n_mat, L.stream_ptr()))
            self._mirror_version = stamp
            self._mirror_epoch += 1
        return self._mirror

    def mark_mirror_fresh(self):
        """libczc itself rewrote the weights AND the mirror in one pass (AdamW, the momentum EMA)."""
        self._mirror_version = self._write_stamp()
        self._mirror_epoch += 1

    def mirror_t(self):
        """The matrix region with every weight matrix transposed (czc_transpose_matrices), for the backward's input-
        gradient products; None when the shape does not tile by 64.  Rebuilt (one launch) whenever the mirror changed."""
        m = self.mirror()
        if self.config.hidden_size % 64 or self.config.intermediate_size % 64:
            return None
        if self._mirror_t is None or self._mirror_t.device != m.device:
            self._mirror_t = torch.empty_like(m)
            self._mirror_t_epoch = -1
        if self._mirror_t_epoch != self._mirror_epoch:
            L.check(L.load().czc_transpose_matrices(C.byref(self._shape), m.data_ptr(), self._mirror_t.data_ptr(), L.stream_ptr()))
            self._mirror_t_epoch = self._mirror_epoch
        return self._mirror_t

    def _workspace(self, B, S, save):
        key = (B, S, int(save))
        ws = self._ws.get(key)
        if ws is None:
            n = int(L.load().czc_encoder_workspace_bytes(C.byref(self._shape), B, S, int(save)))
            # keep at most one workspace per kind; an inference workspace that is large enough is kept as it is
            # (passage encoding pads every batch to its own longest text: the shape changes with every call)
            old = [k for k in self._ws if k[2] == int(save)]
            if not save and old and self._ws[old[0]].numel() >= n:
                return self._ws[old[0]]
            for k in old:
                del self._ws[k]
            ws = torch.empty(n, dtype=torch.uint8, device=self._flat.device)
            self._ws[key] = ws
        return ws

    def _lease_activations(self, B, S, stamp):
        """Activation workspace for a forward whose backward is still to come.  Normally ONE buffer per shape, reused
        every step.  A second differentiable forward of the same encoder before the first backward (the in-batch
        objective encodes queries AND keys with one encoder, src/inbatch.py:64-65) gets a second buffer; with both out,
        the older one is taken over -- its forward was abandoned (no backward came), or its backward will say so."""
        for shape in [k for k in self._ws_pool if k != (B, S)]:   # other shapes: keep only buffers still awaiting a backward
            self._ws_pool[shape] = [e for e in self._ws_pool[shape] if e["out"]]
            if not self._ws_pool[shape]:
                del self._ws_pool[shape]
        pending = sorted(((e["stamp"], k) for k, v in self._ws_pool.items() if k != (B, S) for e in v))
        for stamp, k in pending[:-2]:   # forwards of other shapes never backpropagated: forget all but the two newest
            self._ws_pool[k] = [e for e in self._ws_pool[k] if e["stamp"] != stamp]
            if not self._ws_pool[k]:
                del self._ws_pool[k]
        bufs = self._ws_pool.setdefault((B, S), [])
        if not bufs:
            bufs.append({"ws": self._workspace(B, S, True), "stamp": 0, "out": False})
        pick = next((e for e in bufs if not e["out"]), None)
        if pick is None and len(bufs) < 2:
            pick = {"ws": torch.empty_like(bufs[0]["ws"]), "stamp": 0, "out": False}
            bufs.append(pick)
        if pick is None:
            pick = min(bufs, key=lambda e: e["stamp"])
        pick["stamp"], pick["out"] = stamp, True
        return pick["ws"]

    def _activations_entry(self, ws):
        for bufs in self._ws_pool.values():
            for e in bufs:
                if e["ws"] is ws:
                    return e
        return None

    def _bwd_scratch(self, B, S):
        n = int(L.load().czc_encoder_backward_scratch_bytes(C.byref(self._shape), B, S))
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = torch.empty(n, dtype=torch.uint8, device=self._flat.device)
        return self._scratch

    # ---- dropout stream position.  NOT in the checkpoint (it keeps the reference's five keys): train.py / distill.py put a
    # resumed run at step * update_freq (set_dropout_position), derived from the CURRENT options ----
    def dropout_seed(self, calls=None) -> int:
        """Seed of this encoder's `calls`-th training forward (default: the next one): (--seed, role salt, position)."""
        calls = self._drop_calls + 1 if calls is None else int(calls)
        return (torch.initial_seed() * 2654435761 + calls * 40503 + (self._salt + 1) * 0x9E3779B1) & 0xFFFFFFFF

    def dropout_state(self) -> dict:
        return {"salt": int(self._salt), "calls": int(self._drop_calls)}

    def load_dropout_state(self, state: dict):
        self._salt = int(state.get("salt", self._salt))
        self._drop_calls = int(state.get("calls", self._drop_calls))

    def _pool_code(self):
        p = getattr(self.config, "pooling", "average")
        if p == "average":
            return 0
        if p == "cls":
            return 1
        raise ValueError(f"unknown pooling {p!r}")

    def _shape_for(self, training):
        return self._shape

    # ---- compute -----------------------------------------------------------------------------------
    def encode(self, input_ids, attention_mask, normalize=False, split=None):
        """split = (n, event): passes that nothing differentiates through may run the last layer + pooling in two batch
        ranges and record the czc event once the first n embeddings are complete (czc_encoder_forward, include/czc.h)."""
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, seq]")
        if not input_ids.is_cuda:
            raise RuntimeError("libczc has no CPU path: inputs must be on the GPU")
        ids = input_ids.to(torch.int64).contiguous()
        mask = attention_mask.to(torch.bool).contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._plist)
        if need_grad:
            if self._anchor is None or self._anchor.device != self._flat.device:
                self._anchor = torch.zeros(1, device=self._flat.device, requires_grad=True)
            return _EncoderFn.apply(self._anchor, self, ids, mask, bool(normalize))
        return self._forward_impl(ids, mask, bool(normalize), save=False, split=split)[0]

    def _forward_impl(self, ids, mask, normalize, save, split=None):
        B, S = ids.shape
        self._call += 1
        if self.training:
            self._drop_calls += 1
        ws = self._lease_activations(B, S, self._call) if save else self._workspace(B, S, False)
        emb = torch.empty(B, self.config.hidden_size, dtype=F32, device=ids.device)
        # (--seed, encoder, training forward): the same run draws the same masks, a resumed run is put at the position
        # step * update_freq by its script; query and key encoder never share a stream of draws
        seed = self.dropout_seed(self._drop_calls)
        n_split, ev = (0, None) if (split is None or save) else (int(split[0]), split[1])
        L.check(L.load().czc_encoder_forward(
            C.byref(self._shape), self._flat.data_ptr(), self.mirror().data_ptr(), ids.data_ptr(), mask.data_ptr(),
            B, S, self._pool_code(), int(normalize), int(self.training), seed, int(save), ws.data_ptr(), ws.numel(),
            emb.data_ptr(), n_split, ev, L.stream_ptr()))
        return emb, seed, ws

    def _backward_impl(self, ids, mask, normalize, seed, ws, d_emb, training):
        B, S = ids.shape
        sc = self._bwd_scratch(B, S)
        for hook in self._pre_grad_access:   # a gradient all-reduce of an earlier backward may still be averaging the array
            hook()
        fresh = any(p.grad is None for p in self._plist if p.requires_grad)
        g = self.flat_grads()
        nl = self._shape.num_layers
        step = self.backward_chunk_layers if (self.backward_chunk_layers > 0 and self._range_hooks) else nl
        la = self._n_mat // nl                       # matrix-region elements per layer
        d_emb = d_emb.contiguous()
        mirror = self.mirror()
        mirror_t = self.mirror_t()
        mt_ptr = mirror_t.data_ptr() if mirror_t is not None else None
        hi = nl
        while hi > 0:
            lo = max(0, hi - step)
            L.check(L.load().czc_encoder_backward(
                C.byref(self._shape), self._flat.data_ptr(), mirror.data_ptr(), mt_ptr, ids.data_ptr(), mask.data_ptr(),
                B, S, self._pool_code(), int(normalize), int(training), seed, ws.data_ptr(), ws.numel(), sc.data_ptr(),
                sc.numel(), d_emb.data_ptr(), g.data_ptr(), 0 if fresh else 1, hi, lo, L.stream_ptr()))
            for hook in self._range_hooks:
                hook(self, lo * la, hi * la)          # these layers' weight matrices are final
            hi = lo
        for hook in self._range_hooks:
            hook(self, self._n_mat, self._n)          # vectors (biases, LayerNorm) and embeddings
        if fresh:
            for (name, shape, off), p in zip(self._table, self._plist):
                if p.requires_grad:
                    p.grad = g[off:off + _numel(shape)].view(shape)
        for hook in self._post_backward:
            hook(self)


class _EncoderFn(torch.autograd.Function):
    """emb = encoder(ids, mask); backward writes parameter gradients straight into the flat grad array."""

    @staticmethod
    def forward(ctx, anchor, enc: FlatBertEncoder, ids, mask, normalize):
        emb, seed, ws = enc._forward_impl(ids, mask, normalize, save=True)
        ctx.enc, ctx.ids, ctx.mask, ctx.normalize, ctx.seed, ctx.ws = enc, ids, mask, normalize, seed, ws
        ctx.training = enc.training
        ctx.stamp = enc._call
        return emb

    @staticmethod
    def backward(ctx, d_emb):
        enc = ctx.enc
        entry = enc._activations_entry(ctx.ws)
        if entry is None or entry["stamp"] != ctx.stamp:
            raise RuntimeError("encoder backward after newer forwards of the same shape: activations were overwritten "
                               "(at most two differentiable forwards of one encoder may be pending)")
        try:
            enc._backward_impl(ctx.ids, ctx.mask, ctx.normalize, ctx.seed, ctx.ws, d_emb.to(F32), ctx.training)
        finally:
            entry["out"] = False
        return torch.zeros_like(d_emb[:1, 0]), None, None, None, None


# ----------------------------------------------------------------------------------------------
# InfoNCE head (autograd wrapper over czc_infonce_forward / _backward)
# ----------------------------------------------------------------------------------------------
class _InfoNCEFn(torch.autograd.Function):
    """loss = CE([q.k | q.queue] / T, 0).  When q needs a gradient, dq is computed HERE, in the forward,
    while the queue still holds the negatives the loss was computed on: the reference differentiates
    through `queue.clone()` (src/moco.py:291), and MoCo.forward overwrites queue rows with this step's
    keys right after the loss, long before train.py calls backward()."""

    @staticmethod
    def forward(ctx, q, k, queue_bf16, temperature, label_smoothing, ws, stats_out, want_grad):
        lib = L.load()
        B, Cdim = q.shape
        K = queue_bf16.shape[0]
        qc, kc = q.contiguous(), k.contiguous()
        lse = torch.empty(B, dtype=F32, device=q.device)
        L.check(lib.czc_infonce_forward(qc.data_ptr(), kc.data_ptr(), queue_bf16.data_ptr(), B, Cdim, K,
                                        float(temperature), float(label_smoothing), ws.data_ptr(), ws.numel(),
                                        lse.data_ptr(), stats_out.data_ptr(), L.stream_ptr()))
        if want_grad:
            dq = torch.empty_like(qc)
            L.check(lib.czc_infonce_backward(qc.data_ptr(), kc.data_ptr(), queue_bf16.data_ptr(), B, Cdim, K,
                                             float(temperature), float(label_smoothing), lse.data_ptr(), 1.0,
                                             ws.data_ptr(), ws.numel(), dq.data_ptr(), L.stream_ptr()))
            ctx.save_for_backward(dq)
        ctx.want_grad = want_grad
        return stats_out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if not ctx.want_grad:
            return (None,) * 8
        (dq,) = ctx.saved_tensors
        return dq * g, None, None, None, None, None, None, None


def infonce_loss(q, k, queue_bf16, temperature, label_smoothing, ws, stats_out):
    want = bool(torch.is_grad_enabled() and q.requires_grad)
    return _InfoNCEFn.apply(q, k, queue_bf16, temperature, label_smoothing, ws, stats_out, want)


# ----------------------------------------------------------------------------------------------
# In-batch head (autograd wrapper over czc_inbatch_head): SURVEY §8 row f4, reference src/inbatch.py:71-75
# ----------------------------------------------------------------------------------------------
class _InBatchHeadFn(torch.autograd.Function):
    """loss = CE((q / T) . keys^T, label_offset + arange(B), label_smoothing); queries AND keys are differentiable.  Both
    gradients come out of the same call as the loss (the scores are in the workspace then); backward scales them."""

    @staticmethod
    def forward(ctx, q, keys, label_offset, temperature, label_smoothing, ws, stats_out):
        lib = L.load()
        B, Cdim = q.shape
        N = keys.shape[0]
        qc, kc = q.contiguous(), keys.contiguous()
        need_q, need_k = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dq = torch.empty_like(qc) if need_q else None
        dk = torch.empty_like(kc) if need_k else None
        L.check(lib.czc_inbatch_head(qc.data_ptr(), kc.data_ptr(), B, N, Cdim, int(label_offset), float(temperature),
                                     float(label_smoothing), ws.data_ptr(), ws.numel(), stats_out.data_ptr(),
                                     None if dq is None else dq.data_ptr(), None if dk is None else dk.data_ptr(), L.stream_ptr()))
        ctx.save_for_backward(*[t for t in (dq, dk) if t is not None])
        ctx.have = (need_q, need_k)
        return stats_out[0].clone()

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        dq = saved.pop(0) * g if ctx.have[0] else None
        dk = saved.pop(0) * g if ctx.have[1] else None
        return dq, dk, None, None, None, None, None


def inbatch_loss(q, keys, label_offset, temperature, label_smoothing, stats_out):
    """-> loss (0-d); stats_out f32[8] on the device receives loss, accuracy %, stdq, stdk (include/czc.h)."""
    n = int(L.load().czc_inbatch_workspace_floats(q.shape[0], keys.shape[0]))
    ws = torch.empty(n, dtype=F32, device=q.device)
    return _InBatchHeadFn.apply(q.to(F32), keys.to(F32), label_offset, temperature, label_smoothing, ws, stats_out)
