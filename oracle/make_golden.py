#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Run:  python oracle/make_golden.py            (needs /root/reference; CPU only)

The reference (Ajchler/czech-contriever) is imported from /root/reference and driven piece
by piece, exactly as SURVEY.md §8(c2) lists: `src.contriever.Contriever` (HF BertModel
arithmetic), `MoCo._compute_logits`, `MoCo._momentum_update_key_encoder`,
`MoCoDistill._dequeue_and_enqueue(keys, None)`, `utils.set_optim`/`WarmupLinearScheduler`.
`MoCo.forward` itself cannot run on CPU (`.cuda()` at src/moco.py:326) and
`MoCo._dequeue_and_enqueue` has an arity bug (src/moco.py:274 vs src/dist_utils.py:30),
so the micro-step trace G8 composes the same pieces in the order of src/moco.py:296-353.

Inputs and weights come from oracle.moco_bert_oracle.det_uniform/det_ints/make_params
(a counter hash, independent of any RNG library), so fixtures hold only small inputs and the
reference's OUTPUTS; the large weight sets are regenerated from their seed by the tests.
Nothing from /root/reference is copied: the .npz files contain numbers only.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import moco_bert_oracle as O  # noqa: E402


def import_reference():
    # `src` also names this repo's drop-in package: load the reference's under its own path
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import src.contriever as rc  # noqa
    import src.moco as rm  # noqa
    import src.utils as ru  # noqa
    import src.dist_utils as rd  # noqa
    import src.inbatch as ri  # noqa
    assert rc.__file__.startswith(REF), rc.__file__
    sys.path.remove(REF)
    return rc, rm, ru, rd, ri


def hf_config(cfg: O.BertShape):
    import transformers

    return transformers.BertConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        layer_norm_eps=cfg.layer_norm_eps, hidden_dropout_prob=cfg.hidden_dropout_prob,
        attention_probs_dropout_prob=cfg.attention_probs_dropout_prob, hidden_act="gelu",
    )


def ref_encoder(rc, cfg, params, pooling):
    enc = rc.Contriever(hf_config(cfg), pooling=pooling)
    enc.config.pooling = pooling
    missing = enc.load_state_dict({k: v.clone() for k, v in params.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    # only non-persistent buffers (position_ids/token_type_ids) may be "missing"
    assert all("_ids" in k for k in missing.missing_keys), missing
    enc.eval()
    return enc


def ragged_batch(B, S, seed, vocab, lengths=None):
    ids = O.det_ints((B, S), seed, 3, vocab)
    if lengths is None:
        lengths = [S] * B
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    ids = ids * mask  # pad id 0 (src/data.py:563-573)
    return ids, mask


def save(name, **arrs):
    out = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(out, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {out}  ({os.path.getsize(out)/1e3:.1f} kB)")


TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=64)
TINY16 = dict(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
              max_position_embeddings=64)
BASE1 = dict(vocab_size=1024, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
             max_position_embeddings=512)


def g_encoder(rc, name, shape_kw, seed, B, S, lengths, hidden_stride=1):
    cfg = O.BertShape(**shape_kw)
    P = O.make_params(cfg, seed)
    ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, lengths)
    out = {"ids": ids, "mask": mask, "seed": seed, "shape": np.array(list(shape_kw.values())),
           "param_checksum": np.array([float(sum(t.double().sum() for t in P.values())),
                                       float(sum((t.double() ** 2).sum() for t in P.values()))])}
    with torch.no_grad():
        for pooling in ("average", "cls"):
            enc = ref_encoder(rc, cfg, P, pooling)
            for norm in (False, True):
                out[f"emb_{pooling}_{int(norm)}"] = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        hs = enc.__class__.__mro__[1].forward(enc, input_ids=ids, attention_mask=mask, output_hidden_states=True)
        hidden = torch.stack(list(hs.hidden_states))  # [L+1,B,S,H]
        out["hidden"] = hidden[:, :, ::hidden_stride]
    save(name, **out)


def g_grads(rc, name, shape_kw, seed, B, S, lengths):
    cfg = O.BertShape(**shape_kw)
    P = O.make_params(cfg, seed)
    ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, lengths)
    cot = O.det_uniform((B, cfg.hidden_size), seed + 2)
    out = {"ids": ids, "mask": mask, "seed": seed, "cot": cot}
    for pooling, norm in (("average", False), ("average", True), ("cls", False)):
        enc = ref_encoder(rc, cfg, P, pooling)
        emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        out[f"emb_{tag}"] = emb.detach()
        for n, p in enc.named_parameters():
            out[f"grad_{tag}::{n}"] = p.grad
    save(name, **out)


def bare_moco(rm, cls_name="MoCo"):
    cls = getattr(rm, cls_name)
    m = cls.__new__(cls)
    torch.nn.Module.__init__(m)
    return m


def g_head(rm):
    B, C, K = 8, 768, 512
    q = O.det_uniform((B, C), 31, -0.4, 0.4)
    k = O.det_uniform((B, C), 32, -0.4, 0.4)
    k[:4] = q[:4] * 0.9 + 0.02  # some easy positives so accuracy is neither 0 nor 100
    queue = torch.nn.functional.normalize(O.det_uniform((C, K), 33), dim=0) * 2.0
    m = bare_moco(rm)
    m.register_buffer("queue", queue.clone())
    out = {"q": q, "k": k, "queue_seed": 33}
    for T, ls in ((0.05, 0.0), (0.05, 0.1), (1.0, 0.0)):
        qq = q.clone().requires_grad_(True)
        logits = m._compute_logits(qq, k) / T
        labels = torch.zeros(B, dtype=torch.long)
        loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=ls)
        loss.backward()
        tag = f"T{T}_ls{ls}"
        acc = 100 * (torch.argmax(logits, dim=-1) == labels).float().mean()
        out[f"logits_{tag}"] = logits.detach()
        out[f"loss_{tag}"] = loss.detach()
        out[f"dq_{tag}"] = qq.grad
        out[f"acc_{tag}"] = acc
    out["stdq"] = torch.std(q, dim=0).mean()
    out["stdk"] = torch.std(k, dim=0).mean()
    save("g3_head.npz", **out)


def g_ema(rc, rm):
    cfg = O.BertShape(**TINY)
    Pq = O.make_params(cfg, 41)
    Pk = O.make_params(cfg, 42)
    m = bare_moco(rm)
    m.momentum = 0.9995
    m.encoder_q = ref_encoder(rc, cfg, Pq, "average")
    m.encoder_k = ref_encoder(rc, cfg, Pk, "average")
    out = {}
    with torch.no_grad():
        for step in range(3):
            for i, p in enumerate(m.encoder_q.parameters()):
                p.add_(O.det_uniform(tuple(p.shape), 4300 + 100 * step + i, -1e-2, 1e-2))
            m._momentum_update_key_encoder()
            flat = torch.cat([p.reshape(-1) for p in m.encoder_k.parameters()])
            out[f"k_flat_step{step}"] = flat[::97].clone()
            out[f"k_sum_step{step}"] = flat.double().sum()
    save("g4_ema.npz", **out)


def g_enqueue(rm):
    C, K, n = 32, 512, 64
    m = bare_moco(rm, "MoCoDistill")
    m.queue_size = K
    m.register_buffer("queue", torch.nn.functional.normalize(O.det_uniform((C, K), 51), dim=0))
    m.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
    ptrs, sums = [], []
    for step in range(10):
        keys = O.det_uniform((n, C), 5200 + step)
        m._dequeue_and_enqueue(keys, None)
        ptrs.append(int(m.queue_ptr))
        sums.append(float(m.queue.double().sum()))
    save("g5_enqueue.npz", queue_final=m.queue, ptrs=np.array(ptrs), sums=np.array(sums))


def g_adamw(ru):
    out = {}
    for warm in (-1, 0, 3):
        opt = types.SimpleNamespace(optim="adamw", weight_decay_from_init=False, lr=1e-2, beta1=0.9, beta2=0.98,
                                    eps=1e-6, weight_decay=0.01, warmup_steps=warm, total_steps=10,
                                    lr_min_ratio=0.1, scheduler="linear")
        p = torch.nn.Parameter(O.det_uniform((1000,), 61))
        mod = torch.nn.Module()
        mod.p = p
        optimizer, scheduler = ru.set_optim(opt, mod)
        lrs, ps = [], []
        for step in range(5):
            p.grad = O.det_uniform((1000,), 6200 + step, -0.5, 0.5)
            lrs.append(scheduler.get_last_lr()[0])
            optimizer.step()
            scheduler.step()
            ps.append(p.detach().clone())
        out[f"lrs_w{warm}"] = np.array(lrs)
        out[f"p_w{warm}"] = torch.stack(ps)
        out[f"lambda_w{warm}"] = np.array([scheduler.lr_lambda(s) for s in range(12)])
    save("g6_adamw.npz", **out)


def g_trace(rc, rm, ru, name="g8_moco_trace.npz", norm=False, ls=0.0):
    """5 MoCo micro-steps (update_freq=1), dropout off, tiny encoder: the loss sequence."""
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 71)
    C, K, B, S = cfg.hidden_size, 64, 8, 24
    m = bare_moco(rm)
    m.momentum, m.temperature, m.label_smoothing = 0.99, 0.05, ls
    m.encoder_q = ref_encoder(rc, cfg, P, "average")
    m.encoder_k = ref_encoder(rc, cfg, P, "average")
    for p in m.encoder_k.parameters():
        p.requires_grad = False
    m.register_buffer("queue", torch.nn.functional.normalize(O.det_uniform((C, K), 72), dim=0))
    m.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
    m.queue_size = K
    enq = rm.MoCoDistill._dequeue_and_enqueue
    opt = types.SimpleNamespace(optim="adamw", weight_decay_from_init=False, lr=1e-3, beta1=0.9, beta2=0.98,
                                eps=1e-6, weight_decay=0.01, warmup_steps=2, total_steps=10, lr_min_ratio=0.0,
                                scheduler="linear")
    optimizer, scheduler = ru.set_optim(opt, m)
    losses, accs, stdqs, stdks, ptrs, qs = [], [], [], [], [], []
    for step in range(5):
        lens_q = [S, 17, 3, 24, 9, 24, 12, 20]
        lens_k = [20, S, 5, 11, 24, 7, 24, 16]
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, cfg.vocab_size, lens_q)
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, cfg.vocab_size, lens_k)
        q = m.encoder_q(input_ids=q_tokens, attention_mask=q_mask, normalize=norm)
        with torch.no_grad():
            m._momentum_update_key_encoder()
            k = m.encoder_k(input_ids=k_tokens, attention_mask=k_mask, normalize=norm)
        logits = m._compute_logits(q, k) / m.temperature
        labels = torch.zeros(B, dtype=torch.long)
        loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=ls)
        enq(m, k, None)
        loss.backward()
        optimizer.step()
        scheduler.step()
        m.zero_grad()
        losses.append(float(loss))
        accs.append(float(100 * (logits.argmax(-1) == labels).float().mean()))
        stdqs.append(float(torch.std(q, dim=0).mean()))
        stdks.append(float(torch.std(k, dim=0).mean()))
        ptrs.append(int(m.queue_ptr))
        qs.append(q.detach().clone())
    save(name, losses=np.array(losses), accs=np.array(accs), stdq=np.array(stdqs),
         stdk=np.array(stdks), ptrs=np.array(ptrs), q=torch.stack(qs), queue_final=m.queue,
         wq_probe=m.encoder_q.state_dict()["encoder.layer.1.output.dense.weight"][::7, ::5],
         wk_probe=m.encoder_k.state_dict()["encoder.layer.1.output.dense.weight"][::7, ::5])


def g_inbatch(rc, ri):
    """Reference InBatch.forward (src/inbatch.py:59-90) on a tiny encoder, dropout off: loss, stats and the gradient
    of every encoder parameter (queries AND keys go through the same encoder)."""
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 31)
    q_ids, q_mask = ragged_batch(6, 24, 32, cfg.vocab_size, [24, 17, 5, 9, 24, 12])
    k_ids, k_mask = ragged_batch(6, 20, 33, cfg.vocab_size, [20, 11, 7, 20, 3, 16])
    out = {"q_ids": q_ids, "q_mask": q_mask, "k_ids": k_ids, "k_mask": k_mask, "seed": 31}
    for tag, norm, temp, ls in (("plain", False, 1.0, 0.0), ("norm", True, 0.05, 0.1)):
        enc = ref_encoder(rc, cfg, P, "average")
        opt = argparse.Namespace(norm_doc=norm, norm_query=norm, label_smoothing=ls, temperature=temp)
        m = ri.InBatch(opt, retriever=enc, tokenizer=object())
        loss, stats = m(q_tokens=q_ids, q_mask=q_mask, k_tokens=k_ids, k_mask=k_mask, stats_prefix="train", iter_stats={})
        loss.backward()
        out[f"{tag}_cfg"] = np.array([float(norm), temp, ls])
        out[f"{tag}_loss"] = loss.detach()
        out[f"{tag}_stats"] = np.array([float(stats["train/loss"][0]), float(stats["train/accuracy"][0]),
                                        float(stats["train/stdq"][0]), float(stats["train/stdk"][0])])
        for n, p in enc.named_parameters():
            out[f"{tag}_grad::{n}"] = p.grad
    save("g10_inbatch.npz", **out)


def g_base_grads(rc):
    """bert-base-WIDTH gradients from the reference (H 768, 12 heads, I 3072, one layer, S 256, ragged): every parameter's
    gradient of sum(emb * cot), subsampled (every `stride`-th element of the flattened tensor, <= 4096 values each) plus its
    float64 sum and L2 norm, so the fixture stays small while every tensor is covered."""
    cfg = O.BertShape(**BASE1)
    seed, B, S = 23, 4, 256
    P = O.make_params(cfg, seed)
    ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, [256, 130, 27, 201])
    cot = O.det_uniform((B, cfg.hidden_size), seed + 2)
    out = {"ids": ids, "mask": mask, "seed": seed, "cot": cot}
    for pooling, norm in (("average", False), ("average", True)):
        enc = ref_encoder(rc, cfg, P, pooling)
        emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        out[f"emb_{tag}"] = emb.detach()
        for n, p in enc.named_parameters():
            g = p.grad.reshape(-1)
            stride = max(1, (g.numel() + 4095) // 4096)
            out[f"grad_{tag}::{n}"] = g[::stride].clone()
            out[f"gstat_{tag}::{n}"] = np.array([float(g.double().sum()), float(g.double().norm()), float(stride)])
    save("g12_base_grads.npz", **out)


def reference_function_from_text(path, names):
    """Run chosen top-level definitions of a reference file that cannot be imported as a module (distill.py needs
    clearml / sentence_transformers / beir at import and calls Task.init at module level): parse the file, keep only the
    named `def`s and simple constant assignments, and execute those nodes -- the reference's own code, nothing restated
    and nothing stubbed.  Build container only; nothing of the text is written anywhere."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in names and isinstance(node.value, ast.Constant):
            keep.append(node)
    found = {n.name if isinstance(n, ast.FunctionDef) else n.targets[0].id for n in keep}
    assert found == set(names), (found, names)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def g_distill():
    """Row f2's extra loss from the reference's own `compute_sim_matrix` and DISTILL_LOSS_MULTIPLIER (distill.py:30,44-47),
    composed as its training loop does (distill.py:272-277: mse of the two similarity matrices, times the multiplier):
    student [8, 768], teacher [8, 3584] -> loss and d loss / d student by autograd; a second, bert-batch-sized case
    [64, 768] x [64, 3584] keeps only the loss and a subsample of the gradient."""
    sim, mult = reference_function_from_text(os.path.join(REF, "distill.py"), ["compute_sim_matrix", "DISTILL_LOSS_MULTIPLIER"])
    out = {"multiplier": float(mult)}
    for tag, B in (("b8", 8), ("b64", 64)):
        student = O.det_uniform((B, 768), 1201 + B, -0.5, 0.5).requires_grad_(True)
        teacher = O.det_uniform((B, 3584), 1301 + B, -1.0, 1.0)
        teacher[: B // 2] += 0.3 * teacher[B // 2: 2 * (B // 2)]       # correlated rows: off-diagonal similarities away from 0
        aux = torch.nn.functional.mse_loss(sim(student), sim(teacher)) * mult
        aux.backward()
        out[f"{tag}_student_seed"] = 1201 + B
        out[f"{tag}_teacher"] = teacher if B == 8 else teacher[:, ::16]
        out[f"{tag}_teacher_full_checksum"] = float(teacher.double().sum())
        out[f"{tag}_loss"] = aux.detach()
        out[f"{tag}_dstudent"] = student.grad
        out[f"{tag}_sim_teacher"] = sim(teacher)
    save("g13_distill.npz", **out)


class GivenDropout:
    """Within the block, torch.nn.functional.dropout(x, p, training=True) is x * (the next scale of `scales`): the
    reference's modules (HF BertEmbeddings / BertSelfOutput / BertOutput via nn.Dropout, eager_attention_forward) then run
    their TRAINING-mode arithmetic under a dropout outcome that is given instead of drawn.  torch is patched, not the
    reference; call order per encoder pass: embeddings, then per layer attention probabilities, attention output, FFN
    output (checked by shape)."""

    def __init__(self, scales):
        self.scales = list(scales)
        self.i = 0

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.dropout

        def given(x, p=0.5, training=True, inplace=False):
            assert training and p > 0, (p, training)
            sc = self.scales[self.i]
            assert tuple(sc.shape) == tuple(x.shape), (self.i, tuple(sc.shape), tuple(x.shape))
            self.i += 1
            return x * sc

        F.dropout = given
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.dropout = self._orig
        if exc[0] is None:
            assert self.i == len(self.scales), (self.i, len(self.scales))


def scales_in_call_order(masks, cfg):
    out = [masks["emb"]]
    for l in range(cfg.num_hidden_layers):
        out += [masks[("attn", l)], masks[("h1", l)], masks[("h2", l)]]
    return out


def train_encoder(rc, cfg, P, pooling):
    enc = ref_encoder(rc, cfg, P, pooling)
    enc.config._attn_implementation = "eager"   # the sdpa path draws its dropout inside the fused op: same arithmetic, not patchable
    enc.train()                                 # train.py:246 model.train(): dropout 0.1 / 0.1 on
    return enc


def g_train_masks(rc, rm, ru, ri):
    """g15: the reference in TRAINING mode (dropout on: train.py:246, src/moco.py:306-321) under dropout outcomes taken
    from the kernels' counter RNG (tests/dropout_ref.py restates it in numpy; the fixture stores seeds, not masks):
      tiny_*   tiny encoder, ragged batch: embeddings + every parameter gradient of sum(emb * cot)
      base_*   bert-base-width layer, S 256 ragged: embeddings + subsampled gradients (as g12)
      split*_  key-encoder style no_grad pass whose last layer ran in two batch ranges (czc_encoder_forward split_batch)
      moco_*   3 MoCo micro-steps with BOTH encoders in train mode (query seeds salt 0, key seeds salt 1)
      inb_*    the in-batch objective's forward / backward in train mode (one encoder, two training forwards)"""
    from tests import dropout_ref as D
    out = {}
    # ---- tiny encoder
    cfg = O.BertShape(**TINY)
    seed, B, S, dseed = 11, 5, 24, D.engine_seed(90001, 1, 0)   # = the engine's first training forward under torch.manual_seed(90001), salt 0
    P = O.make_params(cfg, seed)
    ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, [24, 17, 1, 9, 24])
    cot = O.det_uniform((B, cfg.hidden_size), seed + 2)
    out.update(tiny_ids=ids, tiny_mask=mask, tiny_seed=seed, tiny_cot=cot, tiny_dseed=dseed)
    masks = D.encoder_scales(dseed, cfg, B, S)
    for pooling, norm in (("average", False), ("average", True), ("cls", False)):
        enc = train_encoder(rc, cfg, P, pooling)
        with GivenDropout(scales_in_call_order(masks, cfg)):
            emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        out[f"tiny_emb_{tag}"] = emb.detach()
        for n, p in enc.named_parameters():
            out[f"tiny_grad_{tag}::{n}"] = p.grad
    # ---- bert-base width
    cfg = O.BertShape(**BASE1)
    seed, B, S, dseed = 23, 4, 256, D.engine_seed(90002, 1, 0)
    P = O.make_params(cfg, seed)
    ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, [256, 130, 27, 201])
    cot = O.det_uniform((B, cfg.hidden_size), seed + 2)
    out.update(base_ids=ids, base_mask=mask, base_seed=seed, base_cot=cot, base_dseed=dseed)
    masks = D.encoder_scales(dseed, cfg, B, S)
    for pooling, norm in (("average", False), ("average", True)):
        enc = train_encoder(rc, cfg, P, pooling)
        with GivenDropout(scales_in_call_order(masks, cfg)):
            emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        out[f"base_emb_{tag}"] = emb.detach()
        for n, p in enc.named_parameters():
            g = p.grad.reshape(-1)
            stride = max(1, (g.numel() + 4095) // 4096)
            out[f"base_grad_{tag}::{n}"] = g[::stride].clone()
            out[f"base_gstat_{tag}::{n}"] = np.array([float(g.double().sum()), float(g.double().norm()), float(stride)])
    # ---- the last layer in two batch ranges (tiny, 2 layers; and the base-width single layer)
    for tag, shape_kw, seed, B, S, lengths, split, dseed in (
            ("split3", TINY, 13, 8, 24, [24, 17, 1, 9, 24, 20, 3, 24], 3, D.engine_seed(90003, 1, 1)),   # salt 1: the key encoder
            ("split2b", BASE1, 25, 4, 256, [256, 130, 27, 201], 2, D.engine_seed(90004, 1, 1))):
        cfg = O.BertShape(**shape_kw)
        P = O.make_params(cfg, seed)
        ids, mask = ragged_batch(B, S, seed + 1, cfg.vocab_size, lengths)
        masks = D.encoder_scales(dseed, cfg, B, S, split=split)
        enc = train_encoder(rc, cfg, P, "average")
        with torch.no_grad(), GivenDropout(scales_in_call_order(masks, cfg)):
            emb = enc(input_ids=ids, attention_mask=mask, normalize=False)
        out.update({f"{tag}_ids": ids, f"{tag}_mask": mask, f"{tag}_seed": seed, f"{tag}_dseed": dseed,
                    f"{tag}_split": split, f"{tag}_emb": emb})
    # ---- MoCo micro-steps, both encoders in train mode
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 71)
    C, K, B, S, init_seed = cfg.hidden_size, 64, 8, 24, 4242
    m = bare_moco(rm)
    m.momentum, m.temperature, m.label_smoothing = 0.99, 0.05, 0.1   # g9's flags: normalised embeddings, every step has a real gradient
    m.encoder_q = train_encoder(rc, cfg, P, "average")
    m.encoder_k = train_encoder(rc, cfg, P, "average")
    for p in m.encoder_k.parameters():
        p.requires_grad = False
    m.register_buffer("queue", torch.nn.functional.normalize(O.det_uniform((C, K), 72), dim=0))
    m.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
    m.queue_size = K
    enq = rm.MoCoDistill._dequeue_and_enqueue
    opt = types.SimpleNamespace(optim="adamw", weight_decay_from_init=False, lr=1e-3, beta1=0.9, beta2=0.98,
                                eps=1e-6, weight_decay=0.01, warmup_steps=2, total_steps=10, lr_min_ratio=0.0,
                                scheduler="linear")
    optimizer, scheduler = ru.set_optim(opt, m)
    losses, qs, ks = [], [], []
    for step in range(3):
        lens_q = [S, 17, 3, 24, 9, 24, 12, 20]
        lens_k = [20, S, 5, 11, 24, 7, 24, 16]
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, cfg.vocab_size, lens_q)
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, cfg.vocab_size, lens_k)
        qm = D.encoder_scales(D.engine_seed(init_seed, step + 1, 0), cfg, B, S)
        km = D.encoder_scales(D.engine_seed(init_seed, step + 1, 1), cfg, B, S)
        with GivenDropout(scales_in_call_order(qm, cfg) + scales_in_call_order(km, cfg)):
            q = m.encoder_q(input_ids=q_tokens, attention_mask=q_mask, normalize=True)
            with torch.no_grad():
                m._momentum_update_key_encoder()
                k = m.encoder_k(input_ids=k_tokens, attention_mask=k_mask, normalize=True)
        logits = m._compute_logits(q, k) / m.temperature
        loss = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long), label_smoothing=m.label_smoothing)
        enq(m, k, None)
        loss.backward()
        optimizer.step()
        scheduler.step()
        m.zero_grad()
        losses.append(float(loss)); qs.append(q.detach().clone()); ks.append(k.detach().clone())
    out.update(moco_init_seed=init_seed, moco_losses=np.array(losses), moco_q=torch.stack(qs), moco_k=torch.stack(ks),
               moco_queue_final=m.queue,
               moco_wq_probe=m.encoder_q.state_dict()["encoder.layer.1.output.dense.weight"][::7, ::5],
               moco_wk_probe=m.encoder_k.state_dict()["encoder.layer.1.output.dense.weight"][::7, ::5])
    # ---- the in-batch objective (src/inbatch.py:59-90), train mode: ONE encoder embeds queries, then keys -- two training
    #      forwards, i.e. two consecutive positions of its dropout stream (salt 0, calls 1 and 2)
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 31)
    q_ids, q_mask = ragged_batch(6, 24, 32, cfg.vocab_size, [24, 17, 5, 9, 24, 12])
    k_ids, k_mask = ragged_batch(6, 20, 33, cfg.vocab_size, [20, 11, 7, 20, 3, 16])
    inb_seed = 5151
    enc = train_encoder(rc, cfg, P, "average")
    opt_ns = argparse.Namespace(norm_doc=True, norm_query=True, label_smoothing=0.1, temperature=0.05)
    m_in = ri.InBatch(opt_ns, retriever=enc, tokenizer=object())
    m_in.train()
    qm = D.encoder_scales(D.engine_seed(inb_seed, 1, 0), cfg, 6, 24)
    km = D.encoder_scales(D.engine_seed(inb_seed, 2, 0), cfg, 6, 20)
    with GivenDropout(scales_in_call_order(qm, cfg) + scales_in_call_order(km, cfg)):
        loss, stats = m_in(q_tokens=q_ids, q_mask=q_mask, k_tokens=k_ids, k_mask=k_mask, stats_prefix="train", iter_stats={})
    loss.backward()
    out.update(inb_q_ids=q_ids, inb_q_mask=q_mask, inb_k_ids=k_ids, inb_k_mask=k_mask, inb_init_seed=inb_seed, inb_loss=loss.detach(),
               inb_stats=np.array([float(stats["train/loss"][0]), float(stats["train/accuracy"][0]),
                                   float(stats["train/stdq"][0]), float(stats["train/stdk"][0])]))
    for n, p_ in enc.named_parameters():
        out[f"inb_grad::{n}"] = p_.grad
    save("g15_train_masks.npz", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rc, rm, ru, rd, ri = import_reference()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    jobs = {
        "g1": lambda: g_encoder(rc, "g1_tiny_encoder.npz", TINY, 11, 5, 24, [24, 17, 1, 9, 24]),
        "g1b": lambda: g_encoder(rc, "g1b_tiny_d16_encoder.npz", TINY16, 12, 5, 24, [24, 17, 1, 9, 24]),
        "g2a": lambda: g_encoder(rc, "g2_base_layer_s64.npz", BASE1, 21, 4, 64, [64, 40, 1, 27], hidden_stride=4),
        "g2b": lambda: g_encoder(rc, "g2_base_layer_s256.npz", BASE1, 22, 4, 256, [256, 130, 256, 77], hidden_stride=16),
        "g3": lambda: g_head(rm),
        "g4": lambda: g_ema(rc, rm),
        "g5": lambda: g_enqueue(rm),
        "g6": lambda: g_adamw(ru),
        "g7": lambda: g_grads(rc, "g7_tiny_grads.npz", TINY, 11, 5, 24, [24, 17, 1, 9, 24]),
        "g8": lambda: g_trace(rc, rm, ru),
        "g9": lambda: g_trace(rc, rm, ru, "g9_moco_trace_norm.npz", norm=True, ls=0.1),
        "g10": lambda: g_inbatch(rc, ri),
        "g12": lambda: g_base_grads(rc),
        "g13": lambda: g_distill(),
        "g15": lambda: g_train_masks(rc, rm, ru, ri),
    }
    for k, fn in jobs.items():
        if a.only and k not in a.only.split(","):
            continue
        fn()


if __name__ == "__main__":
    main()
