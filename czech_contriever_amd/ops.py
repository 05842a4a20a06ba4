"""Thin tensor-level wrappers over the C ABI (one Python function per czc_* entry point)."""
from __future__ import annotations

import torch

from . import lib as L

BF16, F32 = torch.bfloat16, torch.float32


def _l():
    return L.load()


def gemm(a, b, *, a_layout=0, b_layout=0, epilogue="store", M=None, N=None, K=None, bias=None, aux=None,
         out=None, out2=None, splitk=1):
    """C[M,N] = A.B ; see czc_gemm_bf16.  a: [M,K] (K-major) or [K,M] (MN-major); b: [N,K] or [K,N]."""
    if M is None:
        M = a.shape[0] if a_layout == 0 else a.shape[1]
    if K is None:
        K = a.shape[1] if a_layout == 0 else a.shape[0]
    if N is None:
        N = b.shape[0] if b_layout == 0 else b.shape[1]
    epi = L.EPI[epilogue]
    if out is None:
        if epilogue == "f32":
            out = torch.empty(splitk, M, N, dtype=F32, device=a.device)
        else:
            out = torch.empty(M, N, dtype=BF16, device=a.device)
    if epilogue == "bias_gelu" and out2 is None:
        out2 = torch.empty(M, N, dtype=BF16, device=a.device)
    L.check(_l().czc_gemm_bf16(a_layout, b_layout, epi, M, N, K, L.ptr(a, BF16), a.stride(0), L.ptr(b, BF16),
                               b.stride(0), L.ptr(out), out.stride(-2), L.ptr(bias, F32), L.ptr(aux, BF16),
                               L.ptr(out2), splitk, L.stream_ptr()))
    return (out, out2) if epilogue == "bias_gelu" else out


def reduce_slabs(slabs, dst, accumulate):
    n = dst.numel()
    L.check(_l().czc_reduce_slabs(L.ptr(slabs, F32), slabs.shape[0], n, L.ptr(dst, F32), int(accumulate), L.stream_ptr()))
    return dst


def attention_forward(qkv, mask, B, S, heads, p=0.0, seed=0, site=0, keep_words=False):
    """keep_words=True (seq <= 256, p > 0): also returns the dropout keep decisions as the forward's compare masks
    (int32 words, include/czc.h) for attention_backward(..., keep_words=that tensor)."""
    ctx = torch.empty(B * S, heads * 64, dtype=BF16, device=qkv.device)
    lse = torch.empty(B, heads, S, dtype=F32, device=qkv.device)
    if keep_words:
        words = torch.zeros(_l().czc_attention_keep_words(B, S, heads), dtype=torch.int32, device=qkv.device)
        L.check(_l().czc_attention_forward_keep(L.ptr(qkv, BF16), L.ptr(mask), B, S, heads, p, seed, site, L.ptr(ctx),
                                                L.ptr(lse), L.ptr(words), L.stream_ptr()))
        return ctx, lse, words
    L.check(_l().czc_attention_forward(L.ptr(qkv, BF16), L.ptr(mask), B, S, heads, p, seed, site, L.ptr(ctx),
                                       L.ptr(lse), L.stream_ptr()))
    return ctx, lse


def attention_backward(qkv, mask, ctx, dctx, lse, B, S, heads, p=0.0, seed=0, site=0, d_bias=None, keep_words=None):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, heads, S, dtype=F32, device=qkv.device)
    part = None
    if d_bias is not None:
        part = torch.empty(_l().czc_attention_bias_scratch_floats(B, S, heads), dtype=F32, device=qkv.device)
    if keep_words is not None:
        L.check(_l().czc_attention_backward_keep(L.ptr(qkv, BF16), L.ptr(mask), L.ptr(ctx, BF16), L.ptr(dctx, BF16),
                                                 L.ptr(lse, F32), B, S, heads, p, seed, site, L.ptr(keep_words), L.ptr(delta),
                                                 L.ptr(dqkv), L.ptr(d_bias), L.ptr(part), L.stream_ptr()))
        return dqkv
    L.check(_l().czc_attention_backward(L.ptr(qkv, BF16), L.ptr(mask), L.ptr(ctx, BF16), L.ptr(dctx, BF16),
                                        L.ptr(lse, F32), B, S, heads, p, seed, site, L.ptr(delta), L.ptr(dqkv),
                                        L.ptr(d_bias), L.ptr(part), L.stream_ptr()))
    return dqkv


def residual_ln_forward(y, resid, gamma, beta, eps, p=0.0, seed=0, site=0):
    rows, H = y.shape
    x = torch.empty_like(y)
    stats = torch.empty(rows, 2, dtype=F32, device=y.device)
    L.check(_l().czc_residual_ln_forward(L.ptr(y, BF16), L.ptr(resid, BF16), rows, H, L.ptr(gamma, F32),
                                         L.ptr(beta, F32), eps, p, seed, site, L.ptr(x), L.ptr(stats), L.stream_ptr()))
    return x, stats


def residual_ln_backward(dx, dx2, y, resid, stats, gamma, d_gamma, d_beta, d_bias, p=0.0, seed=0, site=0,
                         use_partials=True):
    rows, H = y.shape
    part = torch.empty(_l().czc_colsum_scratch_floats(H), dtype=F32, device=y.device) if use_partials else None
    dz = torch.empty_like(y)
    dy = torch.empty_like(y) if p > 0 else dz
    L.check(_l().czc_residual_ln_backward(L.ptr(dx, BF16), L.ptr(dx2), L.ptr(y, BF16), L.ptr(resid, BF16),
                                          L.ptr(stats, F32), rows, H, L.ptr(gamma, F32), p, seed, site, L.ptr(dz),
                                          L.ptr(dy), L.ptr(d_gamma), L.ptr(d_beta), L.ptr(d_bias), L.ptr(part), L.stream_ptr()))
    return dz, dy


def embed_ln_forward(ids, word, pos, type0, gamma, beta, eps, p=0.0, seed=0, site=0):
    B, S = ids.shape
    V, H = word.shape
    x = torch.empty(B * S, H, dtype=BF16, device=ids.device)
    stats = torch.empty(B * S, 2, dtype=F32, device=ids.device)
    L.check(_l().czc_embed_ln_forward(L.ptr(ids, torch.int64), B, S, H, V, L.ptr(word, F32), L.ptr(pos, F32),
                                      L.ptr(type0, F32), L.ptr(gamma, F32), L.ptr(beta, F32), eps, p, seed, site,
                                      L.ptr(x), L.ptr(stats), L.stream_ptr()))
    return x, stats


def embed_ln_backward(ids, word, pos, type0, gamma, stats, dx, d_word, d_pos, d_type0, d_gamma, d_beta, p=0.0,
                      seed=0, site=0, use_partials=True):
    B, S = ids.shape
    V, H = word.shape
    part = torch.empty(_l().czc_colsum_scratch_floats(H), dtype=F32, device=ids.device) if use_partials else None
    L.check(_l().czc_embed_ln_backward(L.ptr(ids, torch.int64), B, S, H, V, L.ptr(word, F32), L.ptr(pos, F32),
                                       L.ptr(type0, F32), L.ptr(gamma, F32), L.ptr(stats, F32), p, seed, site,
                                       L.ptr(dx, BF16), L.ptr(d_word), L.ptr(d_pos), L.ptr(d_type0), L.ptr(d_gamma),
                                       L.ptr(d_beta), L.ptr(part), L.stream_ptr()))


def pool_forward(x, mask, B, S, pooling, normalize):
    H = x.shape[-1]
    emb = torch.empty(B, H, dtype=F32, device=x.device)
    raw = torch.empty(B, H, dtype=F32, device=x.device)
    L.check(_l().czc_pool_forward(L.ptr(x, BF16), L.ptr(mask), B, S, H, pooling, int(normalize), L.ptr(emb), L.ptr(raw),
                                  L.stream_ptr()))
    return emb, raw


def pool_backward(d_emb, raw, mask, B, S, pooling, normalize):
    H = d_emb.shape[-1]
    dx = torch.empty(B * S, H, dtype=BF16, device=d_emb.device)
    L.check(_l().czc_pool_backward(L.ptr(d_emb, F32), L.ptr(raw, F32), L.ptr(mask), B, S, H, pooling, int(normalize),
                                   L.ptr(dx), L.stream_ptr()))
    return dx


def colsum(x, col):
    rows, cols = x.shape
    L.check(_l().czc_colsum_bf16(L.ptr(x, BF16), rows, cols, x.stride(0), L.ptr(col, F32), L.stream_ptr()))
    return col


def dropout_mask(seed, site, p, n, device):
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(_l().czc_dropout_mask(seed, site, p, n, L.ptr(keep), L.stream_ptr()))
    return keep


def ema_update(pk, pq, m, pk_bf16=None, n_bf16=0):
    L.check(_l().czc_ema_update(L.ptr(pk, F32), L.ptr(pq, F32), pk.numel(), float(m), float(1.0 - m),
                                L.ptr(pk_bf16), n_bf16, L.stream_ptr()))


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, clip_coef=None, p_bf16=None, n_bf16=0):
    L.check(_l().czc_adamw_step(L.ptr(p, F32), L.ptr(g, F32), L.ptr(m, F32), L.ptr(v, F32), p.numel(), lr, beta1,
                                beta2, eps, wd, step, grad_scale, L.ptr(clip_coef), L.ptr(p_bf16), n_bf16,
                                L.stream_ptr()))


def adamw_ema_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, pk, ema_m, grad_scale=1.0, clip_coef=None, p_bf16=None, n_bf16=0,
                   pk_bf16=None):
    L.check(_l().czc_adamw_ema_step(L.ptr(p, F32), L.ptr(g, F32), L.ptr(m, F32), L.ptr(v, F32), p.numel(), lr, beta1, beta2, eps, wd,
                                    step, grad_scale, L.ptr(clip_coef), L.ptr(p_bf16), n_bf16, L.ptr(pk, F32), L.ptr(pk_bf16),
                                    float(ema_m), float(1.0 - ema_m), L.stream_ptr()))


def sumsq(g, partials, out):
    L.check(_l().czc_sumsq(L.ptr(g, F32), g.numel(), L.ptr(partials, F32), L.ptr(out, F32), L.stream_ptr()))


def clip_coef(sumsq_t, pre_scale, max_norm, coef):
    L.check(_l().czc_clip_coef(L.ptr(sumsq_t, F32), pre_scale, max_norm, L.ptr(coef, F32), L.stream_ptr()))


def clamp_(g, lo, hi):
    L.check(_l().czc_clamp(L.ptr(g, F32), g.numel(), lo, hi, L.stream_ptr()))


def cast_bf16(src, dst):
    L.check(_l().czc_cast_bf16(L.ptr(src, F32), L.ptr(dst, BF16), src.numel(), L.stream_ptr()))


def infonce_workspace(B, C, K, device):
    n = _l().czc_infonce_workspace_bytes(B, C, K)
    return torch.empty(n, dtype=torch.uint8, device=device)


def infonce_forward(q, k, queue_bf16, T, ls, ws):
    B, C = q.shape
    K = queue_bf16.shape[0]
    lse = torch.empty(B, dtype=F32, device=q.device)
    stats = torch.empty(8, dtype=F32, device=q.device)
    L.check(_l().czc_infonce_forward(L.ptr(q, F32), L.ptr(k, F32), L.ptr(queue_bf16, BF16), B, C, K, T, ls, L.ptr(ws),
                                     ws.numel(), L.ptr(lse), L.ptr(stats), L.stream_ptr()))
    return lse, stats


def infonce_backward(q, k, queue_bf16, T, ls, lse, loss_scale, ws):
    B, C = q.shape
    K = queue_bf16.shape[0]
    dq = torch.empty_like(q)
    L.check(_l().czc_infonce_backward(L.ptr(q, F32), L.ptr(k, F32), L.ptr(queue_bf16, BF16), B, C, K, T, ls,
                                      L.ptr(lse, F32), loss_scale, L.ptr(ws), ws.numel(), L.ptr(dq), L.stream_ptr()))
    return dq


def enqueue(keys, queue_bf16, queue_f32, queue_ptr):
    n, C = keys.shape
    K = queue_bf16.shape[0]
    L.check(_l().czc_enqueue(L.ptr(keys, F32), n, C, K, L.ptr(queue_bf16, BF16), L.ptr(queue_f32), L.ptr(queue_ptr, torch.int64),
                             L.stream_ptr()))


def queue_import(queue_f32, queue_bf16):
    C_, K = queue_f32.shape
    L.check(_l().czc_queue_import(L.ptr(queue_f32, F32), C_, K, L.ptr(queue_bf16, BF16), L.stream_ptr()))


class _DistillSimLoss(torch.autograd.Function):
    """mean((Sn Sn^T - Tn Tn^T)^2) with its gradient w.r.t. the student embeddings from the same call (czc_distill_sim_loss)."""

    @staticmethod
    def forward(ctx, student, teacher):
        B, D = student.shape
        Dt = teacher.shape[1]
        n = _l().czc_distill_workspace_floats(B, D, Dt)
        ws = torch.empty(n, dtype=F32, device=student.device)
        loss = torch.empty(1, dtype=F32, device=student.device)
        grad = torch.empty_like(student) if student.requires_grad else None
        L.check(_l().czc_distill_sim_loss(L.ptr(student, F32), L.ptr(teacher, F32), B, D, Dt, L.ptr(ws, F32), n,
                                          L.ptr(loss, F32), L.ptr(grad, F32), L.stream_ptr()))
        ctx.grad = grad
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.grad is None else ctx.grad * g), None


def distill_sim_loss(student, teacher):
    """student fp32 [B, D] (may require grad), teacher fp32 [B, Dt] -> 0-d loss = F.mse_loss(sim(student), sim(teacher)),
    sim(E) = normalize(E) @ normalize(E).T  (reference distill.py:44-47, 272-274)."""
    if student.shape[0] != teacher.shape[0]:
        raise ValueError(f"student {tuple(student.shape)} vs teacher {tuple(teacher.shape)}")
    return _DistillSimLoss.apply(student.contiguous().float(), teacher.detach().contiguous().float())


def inbatch_head(q, keys, label_offset, temperature, label_smoothing=0.0, want_dq=True, want_dkeys=True):
    """-> (stats f32[8]: loss, accuracy %, stdq, stdk; dq or None; dkeys or None)"""
    B, C_ = q.shape
    N = keys.shape[0]
    ws = torch.empty(int(_l().czc_inbatch_workspace_floats(B, N)), dtype=F32, device=q.device)
    stats = torch.empty(8, dtype=F32, device=q.device)
    dq = torch.empty_like(q) if want_dq else None
    dk = torch.empty_like(keys) if want_dkeys else None
    L.check(_l().czc_inbatch_head(L.ptr(q, F32), L.ptr(keys, F32), B, N, C_, int(label_offset), float(temperature),
                                  float(label_smoothing), L.ptr(ws), ws.numel(), L.ptr(stats), L.ptr(dq), L.ptr(dk), L.stream_ptr()))
    return stats, dq, dk


def l2_from_init(p, p0, weight_decay, grads=None, upstream=None, accumulate=True):
    """-> penalty (0-d f32); with `grads`: grads (+)= 2 wd upstream (p - p0)"""
    out = torch.empty(1, dtype=F32, device=p.device)
    partials = torch.empty(4096, dtype=F32, device=p.device)
    L.check(_l().czc_l2_from_init_forward(L.ptr(p, F32), L.ptr(p0, F32), p.numel(), float(weight_decay), L.ptr(partials), L.ptr(out),
                                          L.stream_ptr()))
    if grads is not None:
        L.check(_l().czc_l2_from_init_backward(L.ptr(p, F32), L.ptr(p0, F32), p.numel(), float(weight_decay), L.ptr(upstream),
                                               L.ptr(grads, F32), int(accumulate), L.stream_ptr()))
    return out[0]
