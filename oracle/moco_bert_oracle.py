"""CPU oracle for the MoCo contrastive-training hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain fp32 PyTorch-on-CPU restatement of the arithmetic that the
reference performs on the path named by BASELINE.json's north_star.  It exists so
that the HIP kernels can be checked on the GPU box, where /root/reference does not
exist.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product package (czech_contriever_amd/, src/) never does.

Pinned (see oracle/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py):
every function below is compared against the reference's own modules imported in the
build container (src.contriever.Contriever on transformers' BertModel,
MoCo._compute_logits, MoCo._momentum_update_key_encoder,
MoCoDistill._dequeue_and_enqueue, utils.set_optim / WarmupLinearScheduler).
The reference ships no tests or fixtures of its own, so those captured outputs are
the only pin there is.

Reference lines each function follows (paths into the reference tree; "HF:" is
transformers/models/bert/modeling_bert.py, which holds the encoder arithmetic the
reference inherits through `class Contriever(BertModel)`, src/contriever.py:11-13):

  embeddings()        HF BertEmbeddings.forward           (word+type+pos -> LN eps -> dropout)
  self_attention()    HF BertSelfAttention.forward + eager_attention_forward
  encoder_layer()     HF BertSelfOutput / BertIntermediate / BertOutput
  pool()              src/contriever.py:45-55
  compute_logits()    src/moco.py:289-294
  infonce_loss()      src/moco.py:323-330
  ema_update()        src/moco.py:260-269
  enqueue()           src/moco.py:272-287 (with gather_nograd = identity at world size 1,
                      src/dist_utils.py:31-32; rank-order concat at world size W)
  step_stats()        src/moco.py:340-351
  inbatch_loss()      src/inbatch.py:59-90 (row f4: one encoder for queries and keys, in-batch negatives)
  distill_sim_loss()  distill.py:44-47 (compute_sim_matrix) + :272-274 (mse of student vs teacher similarity matrices);
                      distill.py itself cannot be imported here (clearml at import, Task.init at module level):
                      oracle/make_golden.py compiles the text of its compute_sim_matrix function alone (ast, no
                      stand-ins) and runs it -> tests/golden/g13_distill.npz pins this function
  warmup_linear()     src/utils.py:99-118
  adamw_step()        src/utils.py:137-150 -> torch.optim.AdamW semantics (decoupled decay)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class BertShape:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


# --------------------------------------------------------------------------------------
# deterministic, library-independent parameter / input generator
# --------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _splitmix64(x):
    """numpy uint64 vectorised splitmix64 finaliser (public-domain algorithm)."""
    import numpy as np

    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """Reproducible uniform floats that depend only on (shape, seed): no torch / numpy RNG."""
    import numpy as np

    n = int(math.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x100000001B3) & _M64)
        bits = _splitmix64(idx)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
    out = (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
    return torch.from_numpy(out)


def det_ints(shape, seed: int, lo: int, hi: int) -> torch.Tensor:
    import numpy as np

    n = int(math.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x100000001B3 + 77) & _M64)
        bits = _splitmix64(idx)
    vals = (bits % np.uint64(hi - lo)).astype(np.int64) + lo
    return torch.from_numpy(vals.reshape(shape))


def param_names(cfg: BertShape) -> List[Tuple[str, Tuple[int, ...]]]:
    """HF BertModel(add_pooling_layer=False) parameter names, in `parameters()` order."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    out = [
        ("embeddings.word_embeddings.weight", (cfg.vocab_size, H)),
        ("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H)),
        ("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H)),
        ("embeddings.LayerNorm.weight", (H,)),
        ("embeddings.LayerNorm.bias", (H,)),
    ]
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        out += [
            (p + "attention.self.query.weight", (H, H)),
            (p + "attention.self.query.bias", (H,)),
            (p + "attention.self.key.weight", (H, H)),
            (p + "attention.self.key.bias", (H,)),
            (p + "attention.self.value.weight", (H, H)),
            (p + "attention.self.value.bias", (H,)),
            (p + "attention.output.dense.weight", (H, H)),
            (p + "attention.output.dense.bias", (H,)),
            (p + "attention.output.LayerNorm.weight", (H,)),
            (p + "attention.output.LayerNorm.bias", (H,)),
            (p + "intermediate.dense.weight", (I, H)),
            (p + "intermediate.dense.bias", (I,)),
            (p + "output.dense.weight", (H, I)),
            (p + "output.dense.bias", (H,)),
            (p + "output.LayerNorm.weight", (H,)),
            (p + "output.LayerNorm.bias", (H,)),
        ]
    return out


def make_params(cfg: BertShape, seed: int, scale: float = 0.02 * math.sqrt(3.0)) -> Dict[str, torch.Tensor]:
    """Seeded parameters with std ~0.02 (matrices), non-trivial LN gains/biases and biases."""
    params = {}
    for i, (name, shape) in enumerate(param_names(cfg)):
        s = seed * 1000 + i
        if name.endswith("LayerNorm.weight"):
            params[name] = 1.0 + det_uniform(shape, s, -0.1, 0.1)
        elif name.endswith("LayerNorm.bias") or name.endswith(".bias"):
            params[name] = det_uniform(shape, s, -0.05, 0.05)
        else:
            params[name] = det_uniform(shape, s, -scale, scale)
    return params


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
def _dropout(x, p, training, scale=None):
    """torch.nn.Dropout in training mode is x * keep / (1 - p) with keep ~ Bernoulli(1 - p) (HF BertEmbeddings /
    BertSelfOutput / BertOutput `self.dropout`, eager_attention_forward's `nn.functional.dropout`).  torch's own draws
    cannot be reproduced outside torch, so a caller that wants a GIVEN outcome hands the multiplicative `scale`
    (= keep / (1 - p)) in: pinned by g15, where the reference's modules run in train() mode with
    torch.nn.functional.dropout replaced by the same multiplication (oracle/make_golden.py:g_train_masks)."""
    if scale is not None:
        return x * scale
    return F.dropout(x, p=p, training=training) if (training and p > 0.0) else x


def _m(masks, key):
    return None if masks is None else masks.get(key)


def embeddings(P, ids, cfg: BertShape, training=False, masks=None):
    B, S = ids.shape
    x = P["embeddings.word_embeddings.weight"][ids]
    x = x + P["embeddings.token_type_embeddings.weight"][0][None, None, :]
    x = x + P["embeddings.position_embeddings.weight"][:S][None, :, :]
    x = F.layer_norm(
        x, (cfg.hidden_size,), P["embeddings.LayerNorm.weight"], P["embeddings.LayerNorm.bias"], cfg.layer_norm_eps
    )
    return _dropout(x, cfg.hidden_dropout_prob, training, _m(masks, "emb"))


def self_attention(P, pre, x, key_mask, cfg: BertShape, training=False, pscale=None):
    B, S, H = x.shape
    nh, d = cfg.num_attention_heads, cfg.head_dim

    def proj(n):
        y = x @ P[pre + f"attention.self.{n}.weight"].t() + P[pre + f"attention.self.{n}.bias"]
        return y.view(B, S, nh, d).transpose(1, 2)

    q, k, v = proj("query"), proj("key"), proj("value")
    scores = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    # keys at padded positions are excluded; queries are not (HF create_bidirectional_mask)
    bias = torch.zeros(B, 1, 1, S, dtype=x.dtype)
    bias = bias.masked_fill(~key_mask[:, None, None, :].bool(), torch.finfo(x.dtype).min)
    probs = torch.softmax(scores + bias, dim=-1)
    probs = _dropout(probs, cfg.attention_probs_dropout_prob, training, pscale)
    ctx = (probs @ v).transpose(1, 2).reshape(B, S, H)
    return ctx


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def encoder_layer(P, l, x, key_mask, cfg: BertShape, training=False, masks=None):
    pre = f"encoder.layer.{l}."
    H = cfg.hidden_size
    ctx = self_attention(P, pre, x, key_mask, cfg, training, _m(masks, ("attn", l)))
    a = ctx @ P[pre + "attention.output.dense.weight"].t() + P[pre + "attention.output.dense.bias"]
    a = _dropout(a, cfg.hidden_dropout_prob, training, _m(masks, ("h1", l)))
    x1 = F.layer_norm(
        a + x, (H,), P[pre + "attention.output.LayerNorm.weight"], P[pre + "attention.output.LayerNorm.bias"],
        cfg.layer_norm_eps,
    )
    u = x1 @ P[pre + "intermediate.dense.weight"].t() + P[pre + "intermediate.dense.bias"]
    h = gelu_erf(u)
    o = h @ P[pre + "output.dense.weight"].t() + P[pre + "output.dense.bias"]
    o = _dropout(o, cfg.hidden_dropout_prob, training, _m(masks, ("h2", l)))
    return F.layer_norm(
        o + x1, (H,), P[pre + "output.LayerNorm.weight"], P[pre + "output.LayerNorm.bias"], cfg.layer_norm_eps
    )


def pool(last_hidden, mask, pooling="average", normalize=False):
    m = mask.bool()
    last_hidden = last_hidden.masked_fill(~m[..., None], 0.0)
    if pooling == "average":
        emb = last_hidden.sum(dim=1) / m.sum(dim=1)[..., None]
    elif pooling == "cls":
        emb = last_hidden[:, 0]
    else:
        raise ValueError(pooling)
    if normalize:
        emb = F.normalize(emb, dim=-1)
    return emb


def encoder_forward(P, ids, mask, cfg: BertShape, pooling="average", normalize=False, training=False,
                    return_hidden=False, masks=None):
    """masks: the dropout outcome of a training-mode pass as multiplicative scales keep / (1 - p), keys "emb" [B,S,H],
    ("attn", l) [B,heads,S,S], ("h1", l), ("h2", l) [B,S,H] (tests/dropout_ref.encoder_scales builds them from the
    kernels' counter RNG); a site without an entry falls back to `training` (torch's own draws)."""
    x = embeddings(P, ids, cfg, training, masks)
    hidden = [x]
    for l in range(cfg.num_hidden_layers):
        x = encoder_layer(P, l, x, mask, cfg, training, masks)
        hidden.append(x)
    emb = pool(x, mask, pooling, normalize)
    return (emb, hidden) if return_hidden else emb


# --------------------------------------------------------------------------------------
# MoCo head
# --------------------------------------------------------------------------------------
def compute_logits(q, k, queue):
    l_pos = (q * k).sum(dim=1, keepdim=True)
    l_neg = q @ queue
    return torch.cat([l_pos, l_neg], dim=1)


def infonce_loss(q, k, queue, temperature, label_smoothing=0.0):
    logits = compute_logits(q, k, queue) / temperature
    labels = torch.zeros(q.shape[0], dtype=torch.long)
    return F.cross_entropy(logits, labels, label_smoothing=label_smoothing), logits


def step_stats(logits, q, k):
    acc = 100.0 * (logits.argmax(dim=-1) == 0).float().mean()
    return acc, torch.std(q, dim=0).mean(), torch.std(k, dim=0).mean()


def inbatch_loss(qemb, kemb_all_ranks, rank, local_bsz, temperature, label_smoothing=0.0):
    """In-batch negatives (src/inbatch.py:59-90): scores[i, j] = (q_i / T) . k_j over the keys of ALL ranks
    concatenated in rank order, the positive of local query i is column rank * local_bsz + i."""
    labels = torch.arange(qemb.shape[0]) + rank * local_bsz
    scores = (qemb / temperature) @ kemb_all_ranks.t()
    loss = F.cross_entropy(scores, labels, label_smoothing=label_smoothing)
    acc = 100.0 * (scores.argmax(dim=-1) == labels).float().mean()
    return loss, acc, scores


def distill_sim_loss(student, teacher):
    def sim(e):
        e = torch.nn.functional.normalize(e, p=2, dim=1)
        return e @ e.T
    return torch.nn.functional.mse_loss(sim(student), sim(teacher))


def ema_update(Pk, Pq, momentum):
    for n in Pk:
        Pk[n] = Pk[n] * momentum + Pq[n] * (1.0 - momentum)
    return Pk


def enqueue(queue, ptr, keys_all_ranks):
    """queue [C,K]; keys_all_ranks [W*B, C] already concatenated in rank order."""
    n = keys_all_ranks.shape[0]
    K = queue.shape[1]
    assert K % n == 0, f"{n}, {K}"
    queue[:, ptr:ptr + n] = keys_all_ranks.t()
    return queue, (ptr + n) % K


def init_queue(C, K, seed):
    g = torch.Generator().manual_seed(seed)
    return F.normalize(torch.randn(C, K, generator=g), dim=0)


# --------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------
def warmup_linear(step, warmup, total, ratio):
    if step < warmup:
        return (1 - ratio) * step / float(max(1, warmup))
    return max(0.0, 1.0 + (ratio - 1) * (step - warmup) / float(max(1.0, total - warmup)))


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """One torch.optim.AdamW update (step counts from 1). Returns new (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = m * beta1 + g * (1.0 - beta1)
    v = v * beta2 + g * g * (1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# --------------------------------------------------------------------------------------
# one full micro-step (src/moco.py:296-353 + train.py:263-264,324); dropout off, or on with given outcomes
# --------------------------------------------------------------------------------------
class MoCoState:
    def __init__(self, cfg: BertShape, Pq, queue, momentum, temperature, label_smoothing=0.0,
                 norm_query=False, norm_doc=False, pooling="average", l2_from_init=0.0):
        self.cfg = cfg
        self.l2_from_init = l2_from_init                 # src/moco.py:332-336 (weight_decay_from_init)
        self.P0 = {n: t.clone() for n, t in Pq.items()}
        self.Pq = {n: t.clone().requires_grad_(True) for n, t in Pq.items()}
        self.Pk = {n: t.clone() for n, t in Pq.items()}
        self.queue = queue.clone()
        self.ptr = 0
        self.momentum, self.temperature, self.label_smoothing = momentum, temperature, label_smoothing
        self.norm_query, self.norm_doc, self.pooling = norm_query, norm_doc, pooling
        self.m = {n: torch.zeros_like(t) for n, t in Pq.items()}
        self.v = {n: torch.zeros_like(t) for n, t in Pq.items()}
        self.opt_step = 0

    def forward_backward(self, q_tokens, q_mask, k_tokens, k_mask, training=False, distill=None, q_masks=None,
                         k_masks=None):
        """distill = (teacher_embeddings, weight, multiplier): the student step of distill.py:263-278 — the gradient is
        that of (1 - weight) * contrastive + weight * multiplier * distill_sim_loss(q, teacher).
        q_masks / k_masks: given dropout outcomes of the two encoders (encoder_forward's `masks`): the reference keeps
        BOTH in train mode (train.py:246; src/moco.py:316-317 is a no-op guard after model.train())."""
        cfg = self.cfg
        q = encoder_forward(self.Pq, q_tokens, q_mask, cfg, self.pooling, self.norm_query, training, masks=q_masks)
        with torch.no_grad():
            self.Pk = ema_update(self.Pk, {n: t.detach() for n, t in self.Pq.items()}, self.momentum)
            k = encoder_forward(self.Pk, k_tokens, k_mask, cfg, self.pooling, self.norm_doc, training, masks=k_masks)
        loss, logits = infonce_loss(q, k, self.queue.clone(), self.temperature, self.label_smoothing)
        if self.l2_from_init:
            for n, p in self.Pq.items():
                loss = loss + self.l2_from_init * ((p - self.P0[n]) ** 2).sum()
        with torch.no_grad():
            self.queue, self.ptr = enqueue(self.queue, self.ptr, k)
        extra = {}
        if distill is None:
            loss.backward()
        else:
            teacher, weight, mult = distill
            aux = distill_sim_loss(q, teacher) * mult
            ((1 - weight) * loss + weight * aux).backward()
            extra["distill_loss"] = aux.detach()
        acc, stdq, stdk = step_stats(logits.detach(), q.detach(), k)
        return loss.detach(), q.detach(), k, dict({"accuracy": acc, "stdq": stdq, "stdk": stdk}, **extra)

    def optimizer_step(self, lr, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.01, keep_grads=False):
        self.opt_step += 1
        with torch.no_grad():
            for n, p in self.Pq.items():
                if p.grad is None:
                    continue
                newp, self.m[n], self.v[n] = adamw_step(
                    p.detach(), p.grad, self.m[n], self.v[n], self.opt_step, lr, beta1, beta2, eps, weight_decay)
                p.copy_(newp)
                if not keep_grads:
                    p.grad = None
