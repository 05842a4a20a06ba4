"""The CPU oracle (oracle/moco_bert_oracle.py) against outputs captured from the reference itself
(tests/golden/*.npz, made by oracle/make_golden.py). fp32, dropout off. Tolerances are fp32
round-off for a different but algebraically identical op order."""
import os

import numpy as np
import pytest
import torch

from oracle import moco_bert_oracle as O
from oracle.make_golden import TINY, TINY16, BASE1

ATOL = 2e-5


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.mark.parametrize("name,shape_kw,stride", [
    ("g1_tiny_encoder.npz", TINY, 1), ("g1b_tiny_d16_encoder.npz", TINY16, 1),
    ("g2_base_layer_s64.npz", BASE1, 4), ("g2_base_layer_s256.npz", BASE1, 16)])
def test_encoder_matches_reference(golden_dir, name, shape_kw, stride):
    g = load(golden_dir, name)
    cfg = O.BertShape(**shape_kw)
    P = O.make_params(cfg, int(g["seed"]))
    chk = np.array([float(sum(v.double().sum() for v in P.values())),
                    float(sum((v.double() ** 2).sum() for v in P.values()))])
    np.testing.assert_allclose(chk, g["param_checksum"], rtol=1e-12)
    ids, mask = t(g["ids"]), t(g["mask"])
    with torch.no_grad():
        for pooling in ("average", "cls"):
            for norm in (0, 1):
                emb, hidden = O.encoder_forward(P, ids, mask, cfg, pooling, bool(norm), return_hidden=True)
                ref = t(g[f"emb_{pooling}_{norm}"])
                assert torch.allclose(emb, ref, atol=ATOL, rtol=1e-4), (pooling, norm, (emb - ref).abs().max())
        hid = torch.stack(hidden)[:, :, ::stride]
        assert torch.allclose(hid, t(g["hidden"]), atol=5e-5, rtol=1e-4), (hid - t(g["hidden"])).abs().max()


def test_encoder_grads_match_reference(golden_dir):
    g = load(golden_dir, "g7_tiny_grads.npz")
    cfg = O.BertShape(**TINY)
    ids, mask, cot = t(g["ids"]), t(g["mask"]), t(g["cot"])
    for pooling, norm in (("average", False), ("average", True), ("cls", False)):
        P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, int(g["seed"])).items()}
        emb = O.encoder_forward(P, ids, mask, cfg, pooling, norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        for n, p in P.items():
            ref = t(g[f"grad_{tag}::{n}"])
            scale = max(1e-6, float(ref.abs().max()))
            assert float((p.grad - ref).abs().max()) <= 2e-4 * scale + 1e-6, (tag, n)


def test_head_matches_reference(golden_dir):
    g = load(golden_dir, "g3_head.npz")
    q, k = t(g["q"]), t(g["k"])
    queue = torch.nn.functional.normalize(O.det_uniform((768, 512), int(g["queue_seed"])), dim=0) * 2.0
    for T, ls in ((0.05, 0.0), (0.05, 0.1), (1.0, 0.0)):
        tag = f"T{T}_ls{ls}"
        qq = q.clone().requires_grad_(True)
        loss, logits = O.infonce_loss(qq, k, queue, T, ls)
        loss.backward()
        assert torch.allclose(logits, t(g[f"logits_{tag}"]), atol=1e-3, rtol=1e-5)
        assert abs(float(loss) - float(g[f"loss_{tag}"])) < 1e-4
        assert torch.allclose(qq.grad, t(g[f"dq_{tag}"]), atol=1e-5, rtol=1e-4)
        acc, stdq, stdk = O.step_stats(logits.detach(), q, k)
        assert float(acc) == float(g[f"acc_{tag}"])
    assert abs(float(stdq) - float(g["stdq"])) < 1e-6 and abs(float(stdk) - float(g["stdk"])) < 1e-6


def test_ema_matches_reference(golden_dir):
    g = load(golden_dir, "g4_ema.npz")
    cfg = O.BertShape(**TINY)
    Pq, Pk = O.make_params(cfg, 41), O.make_params(cfg, 42)
    for step in range(3):
        for i, n in enumerate(Pq):
            Pq[n] = Pq[n] + O.det_uniform(tuple(Pq[n].shape), 4300 + 100 * step + i, -1e-2, 1e-2)
        Pk = O.ema_update(Pk, Pq, 0.9995)
        flat = torch.cat([Pk[n].reshape(-1) for n, _ in O.param_names(cfg)])
        assert torch.equal(flat[::97], t(g[f"k_flat_step{step}"]))


def test_enqueue_matches_reference(golden_dir):
    g = load(golden_dir, "g5_enqueue.npz")
    C, K, n = 32, 512, 64
    queue = torch.nn.functional.normalize(O.det_uniform((C, K), 51), dim=0)
    ptr = 0
    for step in range(10):
        queue, ptr = O.enqueue(queue, ptr, O.det_uniform((n, C), 5200 + step))
        assert ptr == int(g["ptrs"][step])
        assert abs(float(queue.double().sum()) - float(g["sums"][step])) < 1e-9
    assert torch.equal(queue, t(g["queue_final"]))


def test_adamw_and_schedule_match_reference(golden_dir):
    g = load(golden_dir, "g6_adamw.npz")
    for warm in (-1, 0, 3):
        lam = [O.warmup_linear(s, warm, 10, 0.1) for s in range(12)]
        np.testing.assert_allclose(lam, g[f"lambda_w{warm}"], rtol=1e-12, atol=0)
        p = O.det_uniform((1000,), 61)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in range(5):
            lr = 1e-2 * O.warmup_linear(step, warm, 10, 0.1)
            assert abs(lr - float(g[f"lrs_w{warm}"][step])) < 1e-15
            grad = O.det_uniform((1000,), 6200 + step, -0.5, 0.5)
            p, m, v = O.adamw_step(p, grad, m, v, step + 1, lr, 0.9, 0.98, 1e-6, 0.01)
            assert torch.allclose(p, t(g[f"p_w{warm}"][step]), atol=2e-7, rtol=1e-6)


@pytest.mark.parametrize("name,norm,ls", [("g8_moco_trace.npz", False, 0.0), ("g9_moco_trace_norm.npz", True, 0.1)])
def test_moco_trace_matches_reference(golden_dir, name, norm, ls):
    from oracle.make_golden import ragged_batch
    g = load(golden_dir, name)
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 71)
    queue = torch.nn.functional.normalize(O.det_uniform((cfg.hidden_size, 64), 72), dim=0)
    st = O.MoCoState(cfg, P, queue, momentum=0.99, temperature=0.05, label_smoothing=ls, norm_query=norm, norm_doc=norm)
    B, S = 8, 24
    for step in range(5):
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, cfg.vocab_size, [S, 17, 3, 24, 9, 24, 12, 20])
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, cfg.vocab_size, [20, S, 5, 11, 24, 7, 24, 16])
        loss, q, k, stats = st.forward_backward(q_tokens, q_mask, k_tokens, k_mask)
        st.optimizer_step(1e-3 * O.warmup_linear(step, 2, 10, 0.0))
        assert abs(float(loss) - float(g["losses"][step])) < 2e-4, (step, float(loss), float(g["losses"][step]))
        assert float(stats["accuracy"]) == float(g["accs"][step])
        assert st.ptr == int(g["ptrs"][step])
        assert torch.allclose(q, t(g["q"][step]), atol=1e-4)
    assert torch.allclose(st.queue, t(g["queue_final"]), atol=1e-4)
    assert torch.allclose(st.Pq["encoder.layer.1.output.dense.weight"].detach()[::7, ::5], t(g["wq_probe"]), atol=1e-5)
    assert torch.allclose(st.Pk["encoder.layer.1.output.dense.weight"][::7, ::5], t(g["wk_probe"]), atol=1e-6)


@pytest.mark.parametrize("tag", ["plain", "norm"])
def test_inbatch_matches_reference(golden_dir, tag):
    """the reference's InBatch.forward (src/inbatch.py:59-90), one encoder for queries and keys."""
    g = load(golden_dir, "g10_inbatch.npz")
    cfg = O.BertShape(**TINY)
    norm, temp, ls = (float(x) for x in g[f"{tag}_cfg"])
    P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, int(g["seed"])).items()}
    q = O.encoder_forward(P, t(g["q_ids"]), t(g["q_mask"]), cfg, "average", bool(norm))
    k = O.encoder_forward(P, t(g["k_ids"]), t(g["k_mask"]), cfg, "average", bool(norm))
    loss, acc, _ = O.inbatch_loss(q, k, 0, k.shape[0], temp, ls)
    loss.backward()
    ref = g[f"{tag}_stats"]
    assert abs(float(loss) - float(ref[0])) < 2e-5 * max(1.0, abs(float(ref[0])))
    assert float(acc) == float(ref[1])
    assert abs(float(torch.std(q, dim=0).mean()) - float(ref[2])) < 1e-5
    assert abs(float(torch.std(k, dim=0).mean()) - float(ref[3])) < 1e-5
    for n, p in P.items():
        r = t(g[f"{tag}_grad::{n}"])
        scale = max(1e-6, float(r.abs().max()))
        assert float((p.grad - r).abs().max()) <= 3e-4 * scale + 1e-6, (tag, n)


def base_grad_views(g, tag, named_grads):
    """(name, subsampled gradient, reference subsample, reference [sum, norm, stride]) for fixture g12"""
    for n, grad in named_grads:
        stat = g[f"gstat_{tag}::{n}"]
        stride = int(stat[2])
        yield n, grad.reshape(-1)[::stride], t(g[f"grad_{tag}::{n}"]), stat


def test_base_width_grads_match_reference(golden_dir):
    """bert-base width (H 768, 12 heads, I 3072, S 256): every parameter gradient of the oracle against the reference's
    (g12: subsample + float64 sum and L2 norm of each tensor)."""
    g = load(golden_dir, "g12_base_grads.npz")
    cfg = O.BertShape(**BASE1)
    ids, mask, cot = t(g["ids"]), t(g["mask"]), t(g["cot"])
    for pooling, norm in (("average", False), ("average", True)):
        P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, int(g["seed"])).items()}
        emb = O.encoder_forward(P, ids, mask, cfg, pooling, norm)
        tag = f"{pooling}_{int(norm)}"
        assert torch.allclose(emb, t(g[f"emb_{tag}"]), atol=ATOL, rtol=1e-4)
        (emb * cot).sum().backward()
        gmax = max(float(np.abs(g[f"grad_{tag}::{n}"]).max()) for n in P)
        for n, sub, ref, stat in base_grad_views(g, tag, [(n, p.grad) for n, p in P.items()]):
            scale = max(1e-3 * gmax, float(ref.abs().max()))
            assert float((sub - ref).abs().max()) <= 5e-4 * scale, (tag, n, float((sub - ref).abs().max()), scale)
            assert abs(float(P[n].grad.double().norm()) - stat[1]) <= 1e-4 * stat[1] + 1e-6 * gmax, (tag, n)


def distill_case(g, tag):
    B = 8 if tag == "b8" else 64
    student = O.det_uniform((B, 768), int(g[f"{tag}_student_seed"]), -0.5, 0.5)
    teacher = O.det_uniform((B, 3584), 1301 + B, -1.0, 1.0)
    teacher[: B // 2] += 0.3 * teacher[B // 2: 2 * (B // 2)]
    assert abs(float(teacher.double().sum()) - float(g[f"{tag}_teacher_full_checksum"])) < 1e-6
    return student, teacher


@pytest.mark.parametrize("tag", ["b8", "b64"])
def test_distill_loss_matches_reference(golden_dir, tag):
    """row f2: oracle.distill_sim_loss x multiplier against the reference's own compute_sim_matrix + mse x 100 (g13)"""
    g = load(golden_dir, "g13_distill.npz")
    student, teacher = distill_case(g, tag)
    student.requires_grad_(True)
    aux = O.distill_sim_loss(student, teacher) * float(g["multiplier"])
    aux.backward()
    assert float(g["multiplier"]) == 100.0
    assert abs(float(aux) - float(g[f"{tag}_loss"])) <= 1e-5 * abs(float(g[f"{tag}_loss"]))
    ref = t(g[f"{tag}_dstudent"])
    assert float((student.grad - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


# ------------------------------------------------------------------------------------------------
# g15: TRAINING mode (dropout on) under given dropout outcomes -- the reference's modules in train() with
# torch.nn.functional.dropout replaced by multiplication with masks from the kernels' counter RNG
# ------------------------------------------------------------------------------------------------
def test_training_mode_tiny_encoder_matches_reference_under_given_masks(golden_dir):
    from tests import dropout_ref as D
    g = load(golden_dir, "g15_train_masks.npz")
    cfg = O.BertShape(**TINY)
    ids, mask, cot = t(g["tiny_ids"]), t(g["tiny_mask"]), t(g["tiny_cot"])
    masks = D.encoder_scales(int(g["tiny_dseed"]), cfg, *ids.shape)
    # the masks really drop: about 10 % zeros at every site
    for key, sc in masks.items():
        z = float((sc == 0).float().mean())
        assert 0.07 < z < 0.13, (key, z)
    for pooling, norm in (("average", False), ("average", True), ("cls", False)):
        P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, int(g["tiny_seed"])).items()}
        emb = O.encoder_forward(P, ids, mask, cfg, pooling, norm, training=True, masks=masks)
        tag = f"{pooling}_{int(norm)}"
        assert torch.allclose(emb, t(g[f"tiny_emb_{tag}"]), atol=ATOL, rtol=1e-4), tag
        (emb * cot).sum().backward()
        for n, p in P.items():
            ref = t(g[f"tiny_grad_{tag}::{n}"])
            scale = max(1e-6, float(ref.abs().max()))
            assert float((p.grad - ref).abs().max()) <= 2e-4 * scale + 1e-6, (tag, n)
    # and it is NOT the eval-mode result (the fixture pins the dropped arithmetic, not a no-op)
    with torch.no_grad():
        ev = O.encoder_forward(O.make_params(cfg, int(g["tiny_seed"])), ids, mask, cfg, "average", False)
    assert float((ev - t(g["tiny_emb_average_0"])).abs().max()) > 1e-2


def test_training_mode_base_width_matches_reference_under_given_masks(golden_dir):
    from tests import dropout_ref as D
    g = load(golden_dir, "g15_train_masks.npz")
    cfg = O.BertShape(**BASE1)
    ids, mask, cot = t(g["base_ids"]), t(g["base_mask"]), t(g["base_cot"])
    masks = D.encoder_scales(int(g["base_dseed"]), cfg, *ids.shape)
    for pooling, norm in (("average", False), ("average", True)):
        P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, int(g["base_seed"])).items()}
        emb = O.encoder_forward(P, ids, mask, cfg, pooling, norm, training=True, masks=masks)
        tag = f"{pooling}_{int(norm)}"
        assert torch.allclose(emb, t(g[f"base_emb_{tag}"]), atol=ATOL, rtol=1e-4), tag
        (emb * cot).sum().backward()
        for n, p in P.items():
            stat = g[f"base_gstat_{tag}::{n}"]
            ref = t(g[f"base_grad_{tag}::{n}"])
            got = p.grad.reshape(-1)[::int(stat[2])]
            scale = max(1e-6, float(ref.abs().max()))
            assert float((got - ref).abs().max()) <= 5e-4 * scale + 1e-6, (tag, n)
            assert abs(float(p.grad.double().norm()) - stat[1]) <= 1e-4 * stat[1] + 1e-7, (tag, n)


@pytest.mark.parametrize("tag,shape_kw", [("split3", TINY), ("split2b", BASE1)])
def test_training_mode_split_last_layer_matches_reference_under_given_masks(golden_dir, tag, shape_kw):
    from tests import dropout_ref as D
    g = load(golden_dir, "g15_train_masks.npz")
    cfg = O.BertShape(**shape_kw)
    ids, mask = t(g[f"{tag}_ids"]), t(g[f"{tag}_mask"])
    P = O.make_params(cfg, int(g[f"{tag}_seed"]))
    masks = D.encoder_scales(int(g[f"{tag}_dseed"]), cfg, *ids.shape, split=int(g[f"{tag}_split"]))
    with torch.no_grad():
        emb = O.encoder_forward(P, ids, mask, cfg, "average", False, training=True, masks=masks)
        whole = O.encoder_forward(P, ids, mask, cfg, "average", False, training=True,
                                  masks=D.encoder_scales(int(g[f"{tag}_dseed"]), cfg, *ids.shape))
    assert torch.allclose(emb, t(g[f"{tag}_emb"]), atol=ATOL, rtol=1e-4)
    sp = int(g[f"{tag}_split"])
    assert float((whole[sp:] - emb[sp:]).abs().max()) > 1e-3      # the second range's draws differ from an unsplit pass ...
    assert torch.allclose(whole[:sp], emb[:sp], atol=1e-6)         # ... the first range's do not


def test_training_mode_moco_steps_match_reference_under_given_masks(golden_dir):
    from oracle.make_golden import ragged_batch
    from tests import dropout_ref as D
    g = load(golden_dir, "g15_train_masks.npz")
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 71)
    queue = torch.nn.functional.normalize(O.det_uniform((cfg.hidden_size, 64), 72), dim=0)
    st = O.MoCoState(cfg, P, queue, momentum=0.99, temperature=0.05, label_smoothing=0.1, norm_query=True, norm_doc=True)
    B, S, init_seed = 8, 24, int(g["moco_init_seed"])
    for step in range(3):
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, cfg.vocab_size, [S, 17, 3, 24, 9, 24, 12, 20])
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, cfg.vocab_size, [20, S, 5, 11, 24, 7, 24, 16])
        qm = D.encoder_scales(D.engine_seed(init_seed, step + 1, 0), cfg, B, S)
        km = D.encoder_scales(D.engine_seed(init_seed, step + 1, 1), cfg, B, S)
        loss, q, k, stats = st.forward_backward(q_tokens, q_mask, k_tokens, k_mask, training=True, q_masks=qm, k_masks=km)
        st.optimizer_step(1e-3 * O.warmup_linear(step, 2, 10, 0.0))
        assert abs(float(loss) - float(g["moco_losses"][step])) < 2e-4, (step, float(loss), float(g["moco_losses"][step]))
        assert torch.allclose(q, t(g["moco_q"][step]), atol=1e-4)
        assert torch.allclose(k, t(g["moco_k"][step]), atol=1e-4)
    assert torch.allclose(st.queue, t(g["moco_queue_final"]), atol=1e-4)
    assert torch.allclose(st.Pq["encoder.layer.1.output.dense.weight"].detach()[::7, ::5], t(g["moco_wq_probe"]), atol=1e-5)
    assert torch.allclose(st.Pk["encoder.layer.1.output.dense.weight"][::7, ::5], t(g["moco_wk_probe"]), atol=1e-6)


def test_training_mode_inbatch_matches_reference_under_given_masks(golden_dir):
    """Row f4 in train mode: the reference's InBatch.forward under given dropout outcomes (one encoder, the query pass at
    position 1 of its stream and the key pass at position 2)."""
    from tests import dropout_ref as D
    g = load(golden_dir, "g15_train_masks.npz")
    cfg = O.BertShape(**TINY)
    P = {n: v.requires_grad_(True) for n, v in O.make_params(cfg, 31).items()}
    s = int(g["inb_init_seed"])
    q_ids, k_ids = t(g["inb_q_ids"]), t(g["inb_k_ids"])
    q = O.encoder_forward(P, q_ids, t(g["inb_q_mask"]), cfg, "average", True, training=True,
                          masks=D.encoder_scales(D.engine_seed(s, 1, 0), cfg, *q_ids.shape))
    k = O.encoder_forward(P, k_ids, t(g["inb_k_mask"]), cfg, "average", True, training=True,
                          masks=D.encoder_scales(D.engine_seed(s, 2, 0), cfg, *k_ids.shape))
    loss, acc, _ = O.inbatch_loss(q, k, 0, k.shape[0], 0.05, 0.1)
    loss.backward()
    ref = g["inb_stats"]
    assert abs(float(loss) - float(ref[0])) < 2e-5 * max(1.0, abs(float(ref[0])))
    assert float(acc) == float(ref[1])
    for n, p in P.items():
        r = t(g[f"inb_grad::{n}"])
        scale = max(1e-6, float(r.abs().max()))
        assert float((p.grad - r).abs().max()) <= 5e-4 * scale + 1e-6, n
