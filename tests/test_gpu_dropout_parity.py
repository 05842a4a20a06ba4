"""TRAINING-mode (dropout on) parity of the assembled path -- the mode bench.py runs and the reference trains in
(/root/reference train.py:246 `model.train()`; src/moco.py:316-317 is a no-op guard, so BOTH encoders drop).

torch's Philox draws cannot be reproduced, so parity is stated under a GIVEN dropout outcome: the kernels' counter RNG
(csrc/czc_common.h, restated in numpy by tests/dropout_ref.py and exported as czc_dropout_mask) fixes which elements
are dropped at every site of a forward; oracle/make_golden.py ran the REFERENCE's own modules in train() mode under
those outcomes (g15: torch.nn.functional.dropout replaced by the multiplication) and the oracle takes the same masks
(`masks=`).  What these tests establish is that the site ids (SITE_EMB / SITE_ATTN(l) / SITE_H1(l) / SITE_H2(l),
csrc/encoder.hip), the element indexing of every kernel, the split-range site shift, the forward -> backward seed
hand-off and the engine's per-forward seeds compose to the arithmetic the reference performs under those masks.
Bar: pooled embeddings cosine >= 1 - 1e-3 per row; gradients as in test_base_width_backward_vs_reference_golden."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moco_bert_oracle as O  # the checker, never the product
from oracle.make_golden import BASE1, TINY, ragged_batch
from tests import dropout_ref as D
from tests.test_gpu_model import _moco_opt, _write_cfg, cos_rows, load, make_encoder, t


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _train_encoder(shape_kw, seed, pooling, gpu, init_seed, salt):
    """An encoder in train() mode whose NEXT training forward draws seed D.engine_seed(init_seed, 1, salt)."""
    torch.manual_seed(init_seed)
    enc, P, cfg_o = make_encoder(shape_kw, seed, pooling, gpu)
    enc.train()
    enc.load_dropout_state({"salt": salt, "calls": 0})
    assert enc.dropout_seed() == D.engine_seed(init_seed, 1, salt)
    return enc, P, cfg_o


def test_device_masks_equal_the_numpy_restatement_at_every_site(gpu):
    """czc_dropout_mask (the kernels' czc_keep) == tests/dropout_ref.keep_mask for the sites and sizes the encoder uses."""
    from czech_contriever_amd import ops
    seed = D.engine_seed(90001, 1, 0)
    for site, n in [(D.SITE_EMB, 5 * 24 * 128), (D.site_attn(0), 5 * 2 * 24 * 24), (D.site_h1(1), 4099),
                    (D.site_h2(11) + D.SPLIT_SHIFT, 3 * 256 * 768), (D.site_attn(11), 2 * 12 * 256 * 256)]:
        got = ops.dropout_mask(seed, site, 0.1, n, gpu).cpu().numpy()
        assert np.array_equal(got, D.keep_mask(seed, site, 0.1, n)), (site, n)


@pytest.mark.parametrize("pooling,norm", [("average", False), ("average", True), ("cls", False)])
def test_training_mode_tiny_encoder_vs_reference_golden(golden_dir, gpu, pooling, norm):
    g = load(golden_dir, "g15_train_masks.npz")
    ids, mask, cot = t(g["tiny_ids"]).to(gpu), t(g["tiny_mask"]).to(gpu), t(g["tiny_cot"]).to(gpu)
    enc, _, _ = _train_encoder(TINY, int(g["tiny_seed"]), pooling, gpu, 90001, 0)
    assert enc.dropout_seed() == int(g["tiny_dseed"])
    emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
    tag = f"{pooling}_{int(norm)}"
    ref = t(g[f"tiny_emb_{tag}"])
    assert float(cos_rows(emb.detach().cpu(), ref).min()) >= 1 - 1e-3, cos_rows(emb.detach().cpu(), ref)
    (emb * cot).sum().backward()
    gmax = max(float(np.abs(g[f"tiny_grad_{tag}::{n}"]).max()) for n, _ in enc.named_parameters())
    worst = []
    for n, p in enc.named_parameters():
        r = t(g[f"tiny_grad_{tag}::{n}"])
        scale = max(1e-2 * gmax, float(r.abs().max()))
        err = float((p.grad.cpu() - r).abs().max()) / scale
        worst.append((err, n))
        assert err <= 4e-2, (tag, n, err, scale)
    c = torch.nn.functional.cosine_similarity(
        torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double(),
        torch.cat([t(g[f"tiny_grad_{tag}::{n}"]).reshape(-1) for n, _ in enc.named_parameters()]).double(), dim=0)
    assert float(c) >= 1 - 2e-3, (tag, float(c), sorted(worst)[-3:])
    # the eval-mode arithmetic is far from this fixture: the comparison above cannot pass with dropout ignored or misplaced
    enc.eval()
    with torch.no_grad():
        ev = enc(input_ids=ids, attention_mask=mask, normalize=norm).cpu()
    assert float(cos_rows(ev, ref).min()) < 1 - 3e-3


@pytest.mark.parametrize("pooling,norm", [("average", False), ("average", True)])
def test_training_mode_base_width_vs_reference_golden(golden_dir, gpu, pooling, norm):
    """H 768, 12 heads, I 3072, S 256 ragged, dropout 0.1 / 0.1: embeddings and every gradient against the reference."""
    g = load(golden_dir, "g15_train_masks.npz")
    ids, mask, cot = t(g["base_ids"]).to(gpu), t(g["base_mask"]).to(gpu), t(g["base_cot"]).to(gpu)
    enc, _, _ = _train_encoder(BASE1, int(g["base_seed"]), pooling, gpu, 90002, 0)
    assert enc.dropout_seed() == int(g["base_dseed"])
    emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
    tag = f"{pooling}_{int(norm)}"
    assert float(cos_rows(emb.detach().cpu(), t(g[f"base_emb_{tag}"])).min()) >= 1 - 1e-3
    (emb * cot).sum().backward()
    gmax = max(float(np.abs(g[f"base_grad_{tag}::{n}"]).max()) for n, _ in enc.named_parameters())
    num = da = db = 0.0
    for n, p in enc.named_parameters():
        stat = g[f"base_gstat_{tag}::{n}"]
        ref = t(g[f"base_grad_{tag}::{n}"]).double()
        got = p.grad.detach().reshape(-1)[::int(stat[2])].cpu().double()
        scale = max(1e-2 * gmax, float(ref.abs().max()))
        assert float((got - ref).abs().max()) <= 5e-2 * scale, (tag, n, float((got - ref).abs().max()), scale)
        nrm = float(p.grad.detach().double().norm())
        assert abs(nrm - stat[1]) <= 3e-2 * stat[1] + 1e-3 * gmax, (tag, n, nrm, stat[1])
        num += float((got * ref).sum()); da += float((got * got).sum()); db += float((ref * ref).sum())
    assert num / (da ** 0.5 * db ** 0.5) >= 1 - 2e-3, (tag, num / (da ** 0.5 * db ** 0.5))


@pytest.mark.parametrize("tag,shape_kw,init_seed", [("split3", TINY, 90003), ("split2b", BASE1, 90004)])
def test_training_mode_split_key_range_vs_reference_golden(golden_dir, gpu, tag, shape_kw, init_seed):
    """The key encoder under data parallelism: last layer + pooling in two batch ranges (czc_encoder_forward split_batch),
    dropout on: the second range's sites are shifted and its draws are indexed from its own first row."""
    g = load(golden_dir, "g15_train_masks.npz")
    ids, mask = t(g[f"{tag}_ids"]).to(gpu), t(g[f"{tag}_mask"]).to(gpu)
    enc, _, _ = _train_encoder(shape_kw, int(g[f"{tag}_seed"]), "average", gpu, init_seed, 1)
    assert enc.dropout_seed() == int(g[f"{tag}_dseed"])
    with torch.no_grad():
        emb = enc.encode(ids, mask, normalize=False, split=(int(g[f"{tag}_split"]), None)).cpu()
    assert float(cos_rows(emb, t(g[f"{tag}_emb"])).min()) >= 1 - 1e-3, cos_rows(emb, t(g[f"{tag}_emb"]))
    # an UNSPLIT pass at the same seed is a different outcome for the second range (and the same for the first)
    enc.load_dropout_state({"calls": 0})
    with torch.no_grad():
        whole = enc.encode(ids, mask, normalize=False).cpu()
    sp = int(g[f"{tag}_split"])
    ref = t(g[f"{tag}_emb"])
    assert torch.equal(whole[:sp], emb[:sp])
    # (pooling averages the last layer's rows, so a wrong outcome there moves a row's cosine by only a few 1e-4: the
    #  discriminating statement is the distance itself -- the right masks land several times closer than the wrong ones)
    e_right = float((emb[sp:] - ref[sp:]).abs().max()), float((1 - cos_rows(emb[sp:], ref[sp:])).max())
    e_wrong = float((whole[sp:] - ref[sp:]).abs().max()), float((1 - cos_rows(whole[sp:], ref[sp:])).max())
    print(tag, "split range: right masks", e_right, "unsplit masks", e_wrong)
    assert e_right[0] < 0.35 * e_wrong[0] and e_right[1] < 0.2 * e_wrong[1], (e_right, e_wrong)


def test_training_mode_moco_steps_vs_reference_trace(golden_dir, gpu, tmp_path):
    """3 full MoCo micro-steps with dropout ON in both encoders (query salt 0, key salt 1; the engine's own per-step
    seeds) against the trace the reference's pieces produced under the same outcomes (g15 moco_*; flags of g9)."""
    from src import moco, utils
    g = load(golden_dir, "g15_train_masks.npz")
    torch.manual_seed(int(g["moco_init_seed"]))
    mdir = _write_cfg(tmp_path / "tiny", TINY)      # HF defaults: dropout 0.1 / 0.1
    opt = _moco_opt(mdir, norm_query=True, norm_doc=True, label_smoothing=0.1)
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**TINY)
    P = O.make_params(cfg_o, 71)
    model.encoder_q.load_state_dict(P)
    model.encoder_k.load_state_dict(P)
    with torch.no_grad():
        model.queue.copy_(torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0))
    model = model.cuda()
    model.train()
    assert model.encoder_q.training and model.encoder_k.training
    assert model.encoder_q.config.hidden_dropout_prob == 0.1 and model.encoder_q.config.attention_probs_dropout_prob == 0.1
    optimizer, scheduler = utils.set_optim(opt, model)
    st = O.MoCoState(cfg_o, P, torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0), momentum=0.99,
                     temperature=0.05, label_smoothing=0.1, norm_query=True, norm_doc=True)
    B, S = 8, 24
    for step in range(3):
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, 512, [S, 17, 3, 24, 9, 24, 12, 20])
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, 512, [20, S, 5, 11, 24, 7, 24, 16])
        sq, sk = model.encoder_q.dropout_seed(), model.encoder_k.dropout_seed()
        assert sq == D.engine_seed(int(g["moco_init_seed"]), step + 1, 0) and sk == D.engine_seed(int(g["moco_init_seed"]), step + 1, 1)
        loss, stats = model(q_tokens.to(gpu), q_mask.to(gpu), k_tokens.to(gpu), k_mask.to(gpu), stats_prefix="train")
        loss.backward()
        # the oracle, same outcomes, stepping beside the device: its gradient vs the device's, every step
        lo, qo, ko, _ = st.forward_backward(q_tokens, q_mask, k_tokens, k_mask, training=True,
                                            q_masks=D.encoder_scales(sq, cfg_o, B, S), k_masks=D.encoder_scales(sk, cfg_o, B, S))
        # (the flat gradient array's layout is libczc's, not HF order: compare tensor by tensor)
        num = da = db = 0.0
        for n, p in model.encoder_q.named_parameters():
            a, b = p.grad.detach().cpu().double().reshape(-1), st.Pq[n].grad.double().reshape(-1)
            num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
        assert num / (da ** 0.5 * db ** 0.5) > 0.995, (step, num / (da ** 0.5 * db ** 0.5))
        optimizer.step(); scheduler.step(); model.zero_grad()
        st.optimizer_step(1e-3 * O.warmup_linear(step, 2, 10, 0.0))
        ref_loss = float(g["moco_losses"][step])
        assert abs(float(loss) - ref_loss) < 1e-2 * max(1.0, abs(ref_loss)), (step, float(loss), ref_loss)
        assert abs(float(lo) - ref_loss) < 2e-4
    qf = model.queue.cpu()
    assert float((qf - t(g["moco_queue_final"])).abs().max()) < 2e-2
    # the keys that were enqueued are the dropped-out key encoder's outputs: rows of the final queue vs the reference's k
    assert float(cos_rows(qf.t()[16:24], t(g["moco_k"][2])).min()) >= 1 - 1e-3
    wq = model.encoder_q.state_dict()["encoder.layer.1.output.dense.weight"].cpu()[::7, ::5]
    wk = model.encoder_k.state_dict()["encoder.layer.1.output.dense.weight"].cpu()[::7, ::5]
    assert float((wq - t(g["moco_wq_probe"])).abs().max()) < 2.5e-3
    w0 = P["encoder.layer.1.output.dense.weight"][::7, ::5]
    dref, dgot = (t(g["moco_wq_probe"]) - w0).double().reshape(-1), (wq - w0).double().reshape(-1)
    assert float(torch.nn.functional.cosine_similarity(dref, dgot, dim=0)) > 0.9
    assert float((wk - t(g["moco_wk_probe"])).abs().max()) < 1e-4


def test_training_mode_full_depth_vs_oracle(gpu):
    """The bench's shapes (12 layers, H 768, I 3072, 12 heads, S 256) at B = 8, ragged, dropout 0.1 / 0.1 at all 37 sites:
    embeddings and every parameter gradient of the HIP bf16 pass against the fp32 oracle under the same outcomes."""
    shape_kw = dict(vocab_size=2048, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, max_position_embeddings=512)
    enc, P, cfg_o = _train_encoder(shape_kw, 77, "average", gpu, 31337, 0)
    B, S = 8, 256
    ids, mask = ragged_batch(B, S, 78, 2048, [256, 130, 27, 256, 77, 200, 1, 64])
    cot = O.det_uniform((B, 768), 79)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    masks = D.encoder_scales(enc.dropout_seed(), cfg_o, B, S)
    Pg = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ref = O.encoder_forward(Pg, ids, mask, cfg_o, "average", False, training=True, masks=masks)
    (ref * cot).sum().backward()
    emb = enc(input_ids=ids.to(gpu), attention_mask=mask.to(gpu))
    assert float(cos_rows(emb.detach().cpu(), ref.detach()).min()) >= 1 - 1e-3
    (emb * cot.to(gpu)).sum().backward()
    got = torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double()
    want = torch.cat([Pg[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
    c = float(torch.nn.functional.cosine_similarity(got, want, dim=0))
    assert c >= 1 - 3e-3, c
    assert abs(float(got.norm()) / float(want.norm()) - 1) < 3e-2
    worst = min((float(torch.nn.functional.cosine_similarity(p.grad.reshape(-1).cpu().double(), Pg[n].grad.reshape(-1).double(), dim=0)), n)
                for n, p in enc.named_parameters() if n.endswith("dense.weight") or n.endswith("query.weight")
                or n.endswith("key.weight") or n.endswith("value.weight"))
    assert worst[0] >= 0.99, worst


def test_training_mode_inbatch_vs_reference_golden(golden_dir, gpu, tmp_path):
    """Row f4 with dropout on: the in-batch objective (one encoder, two training forwards: positions 1 and 2 of its dropout
    stream) against the reference's InBatch.forward under the same outcomes (g15 inb_*): loss, stats, every gradient."""
    from src import inbatch
    g = load(golden_dir, "g15_train_masks.npz")
    torch.manual_seed(int(g["inb_init_seed"]))
    mdir = _write_cfg(tmp_path / "tiny", TINY)
    opt = _moco_opt(mdir, norm_query=True, norm_doc=True, label_smoothing=0.1, temperature=0.05)
    model = inbatch.InBatch(opt)
    P = O.make_params(O.BertShape(**TINY), 31)
    model.encoder.load_state_dict(P)
    model = model.cuda()
    model.train()
    assert model.encoder.dropout_seed() == D.engine_seed(int(g["inb_init_seed"]), 1, 0)
    loss, stats = model(t(g["inb_q_ids"]).to(gpu), t(g["inb_q_mask"]).to(gpu), t(g["inb_k_ids"]).to(gpu), t(g["inb_k_mask"]).to(gpu),
                        stats_prefix="train", iter_stats={})
    loss.backward()
    ref = g["inb_stats"]
    assert abs(float(loss) - float(ref[0])) < 1e-2 * max(1.0, abs(float(ref[0]))), (float(loss), float(ref[0]))
    assert abs(float(stats["train/accuracy"][0]) - float(ref[1])) <= 100.0 / 6 + 1e-3     # B = 6: one near-tie row at most
    assert abs(float(stats["train/stdq"][0]) - float(ref[2])) < 5e-3 and abs(float(stats["train/stdk"][0]) - float(ref[3])) < 5e-3
    gmax = max(float(np.abs(g[f"inb_grad::{n}"]).max()) for n in P)
    num = da = db = 0.0
    for n, p in model.encoder.named_parameters():
        r = t(g[f"inb_grad::{n}"]).double()
        got = p.grad.detach().cpu().double()
        scale = max(float(r.abs().max()), 0.01 * gmax)
        assert float((got - r).abs().max()) <= 0.05 * scale, (n, float((got - r).abs().max()), scale)
        num += float((got * r).sum()); da += float((got * got).sum()); db += float((r * r).sum())
    assert num / (da ** 0.5 * db ** 0.5) > 0.998


def test_training_mode_distill_student_step_vs_oracle(gpu, tmp_path):
    """Row f2 with dropout on: the student's step of distill.py (MoCoDistill + the similarity-matrix loss against teacher
    embeddings) against MoCoState.forward_backward(distill=..., q_masks, k_masks) under the same outcomes."""
    import distill
    from src import moco, utils
    torch.manual_seed(777)
    mdir = _write_cfg(tmp_path / "tiny", TINY)
    opt = _moco_opt(mdir, norm_query=True, norm_doc=True, label_smoothing=0.0, temperature=0.05, distill_weight=0.5)
    model = moco.MoCoDistill(opt)
    cfg_o = O.BertShape(**TINY)
    P = O.make_params(cfg_o, 71)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    queue = torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.cuda().train()
    st = O.MoCoState(cfg_o, P, queue, momentum=0.99, temperature=0.05, norm_query=True, norm_doc=True)
    B, S = 8, 24
    qt, qm = ragged_batch(B, S, 7300, 512, [S, 17, 3, 24, 9, 24, 12, 20])
    kt, km = ragged_batch(B, S, 7400, 512, [20, S, 5, 11, 24, 7, 24, 16])
    teacher = O.det_uniform((B, 48), 99)
    sq, sk = model.encoder_q.dropout_seed(), model.encoder_k.dropout_seed()
    loss, q_emb, _ = model(qt.to(gpu), qm.to(gpu), kt.to(gpu), km.to(gpu), process_group=None, stats_prefix="train", iter_stats={})
    aux = distill.distill_loss(q_emb, teacher.to(gpu))
    (0.5 * loss + 0.5 * aux).backward()
    lo, qo, ko, extra = st.forward_backward(qt, qm, kt, km, training=True, distill=(teacher, 0.5, distill.DISTILL_LOSS_MULTIPLIER),
                                            q_masks=D.encoder_scales(sq, cfg_o, B, S), k_masks=D.encoder_scales(sk, cfg_o, B, S))
    assert abs(float(loss) - float(lo)) < 1e-2 * max(1.0, abs(float(lo))), (float(loss), float(lo))
    assert abs(float(aux) - float(extra["distill_loss"])) < 1e-2 * max(1.0, abs(float(extra["distill_loss"])))
    assert float(cos_rows(q_emb.detach().cpu(), qo).min()) >= 1 - 1e-3
    num = da = db = 0.0
    for n, p in model.encoder_q.named_parameters():
        a, b = p.grad.detach().cpu().double().reshape(-1), st.Pq[n].grad.double().reshape(-1)
        num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
    assert num / (da ** 0.5 * db ** 0.5) > 0.995, num / (da ** 0.5 * db ** 0.5)


def test_training_mode_config1_sized_steps_vs_oracle(gpu, tmp_path):
    """BASELINE.json configs[0] at its stated sizes (bert-base-uncased shape, chunk_length 64, per_gpu_batch_size 4,
    queue_size 512) in the mode train.py runs it -- dropout 0.1 on in both encoders: three micro-steps with AdamW against the
    oracle stepping under the same dropout outcomes."""
    from src import moco, utils
    from tests.test_gpu_train import BERT_BASE_UNCASED
    torch.manual_seed(2024)
    mdir = str(tmp_path / "bbu"); os.makedirs(mdir)
    import json
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(BERT_BASE_UNCASED, f)
    K = 512
    opt = _moco_opt(mdir, queue_size=K, momentum=0.9995, temperature=0.05, projection_size=768, norm_query=True, norm_doc=True,
                    lr=1e-4, warmup_steps=0, total_steps=100, lr_min_ratio=1.0)
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**BERT_BASE_UNCASED)
    P = O.make_params(cfg_o, 93)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    queue = O.init_queue(768, K, 0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.to(gpu).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    st = O.MoCoState(cfg_o, P, queue, momentum=0.9995, temperature=0.05, norm_query=True, norm_doc=True)
    B, S = 4, 64
    for step in range(3):
        qt, qm = ragged_batch(B, S, 9700 + step, 30522, [64, 40, 9, 27])
        kt, km = ragged_batch(B, S, 9800 + step, 30522, [33, 64, 64, 12])
        sq, sk = model.encoder_q.dropout_seed(), model.encoder_k.dropout_seed()
        loss, stats = model(qt.to(gpu), qm.to(gpu), kt.to(gpu), km.to(gpu), stats_prefix="train", iter_stats={})
        loss.backward()
        lo, qo, ko, rs = st.forward_backward(qt, qm, kt, km, training=True, q_masks=D.encoder_scales(sq, cfg_o, B, S),
                                             k_masks=D.encoder_scales(sk, cfg_o, B, S))
        g_dev = torch.cat([p.grad.reshape(-1) for p in model.encoder_q.parameters()]).cpu().double()
        g_ref = torch.cat([st.Pq[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
        c = float(torch.nn.functional.cosine_similarity(g_dev, g_ref, dim=0))
        assert abs(float(loss) - float(lo)) < 2e-2 * max(1.0, abs(float(lo))), (step, float(loss), float(lo))
        assert c > 0.99, (step, c)
        optimizer.step(); scheduler.step(); model.zero_grad()
        st.optimizer_step(1e-4)
        assert int(model.queue_ptr) == st.ptr
    assert float(cos_rows(model.queue.cpu().t()[:12], st.queue.t()[:12]).min()) >= 1 - 1e-3
