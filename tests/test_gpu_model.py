"""GPU parity of the assembled path (encoder forward / backward, MoCo step) through the drop-in
Python surface (src.contriever.Contriever, src.moco.MoCo) -> C ABI -> HIP, against
  (a) golden vectors captured from the reference (tests/golden, made by oracle/make_golden.py), and
  (b) the CPU oracle on the same seeded inputs at sizes it finishes in seconds.
Tolerance for pooled embeddings is the north-star bar: cosine >= 1 - 1e-3 per row (bf16 compute
vs the fp32 reference); absolute tolerances are written at each check."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moco_bert_oracle as O  # the checker, never the product
from oracle.make_golden import BASE1, TINY, ragged_batch


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def t(x):
    return torch.from_numpy(np.asarray(x))


def make_encoder(shape_kw, seed, pooling, gpu):
    from src.contriever import Contriever, bert_config
    cfg_o = O.BertShape(**shape_kw)
    cfg = bert_config(vocab_size=cfg_o.vocab_size, hidden_size=cfg_o.hidden_size, num_hidden_layers=cfg_o.num_hidden_layers,
                      num_attention_heads=cfg_o.num_attention_heads, intermediate_size=cfg_o.intermediate_size,
                      max_position_embeddings=cfg_o.max_position_embeddings)
    enc = Contriever(cfg, pooling=pooling)
    P = O.make_params(cfg_o, seed)
    missing = enc.load_state_dict(P, strict=True)
    enc = enc.cuda()
    return enc, P, cfg_o


def cos_rows(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)


@pytest.mark.parametrize("name,shape_kw", [("g1_tiny_encoder.npz", TINY), ("g2_base_layer_s64.npz", BASE1),
                                           ("g2_base_layer_s256.npz", BASE1)])
def test_encoder_forward_vs_reference_golden(golden_dir, gpu, name, shape_kw):
    g = load(golden_dir, name)
    ids, mask = t(g["ids"]).to(gpu), t(g["mask"]).to(gpu)
    for pooling in ("average", "cls"):
        enc, _, _ = make_encoder(shape_kw, int(g["seed"]), pooling, gpu)
        enc.eval()
        for norm in (0, 1):
            with torch.no_grad():
                emb = enc(input_ids=ids, attention_mask=mask, normalize=bool(norm)).cpu()
            ref = t(g[f"emb_{pooling}_{norm}"])
            c = cos_rows(emb, ref)
            assert float(c.min()) >= 1 - 1e-3, (pooling, norm, c)
            scale = float(ref.abs().max())
            assert float((emb - ref).abs().max()) <= 3e-2 * scale, (pooling, norm, float((emb - ref).abs().max()), scale)


def test_encoder_state_dict_roundtrip_names(gpu):
    enc, P, cfg_o = make_encoder(TINY, 5, "average", gpu)
    sd = enc.state_dict()
    assert list(sd.keys()) == [n for n, _ in O.param_names(cfg_o)]
    for n, v in P.items():
        assert torch.equal(sd[n].cpu(), v), n


def test_encoder_backward_vs_reference_golden(golden_dir, gpu):
    g = load(golden_dir, "g7_tiny_grads.npz")
    ids, mask, cot = t(g["ids"]).to(gpu), t(g["mask"]).to(gpu), t(g["cot"]).to(gpu)
    for pooling, norm in (("average", False), ("average", True), ("cls", False)):
        enc, _, cfg_o = make_encoder(TINY, int(g["seed"]), pooling, gpu)
        enc.eval()  # dropout off, gradients on
        emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        (emb * cot).sum().backward()
        tag = f"{pooling}_{int(norm)}"
        worst = []
        gmax = max(float(t(g[f"grad_{tag}::{n}"]).abs().max()) for n, _ in enc.named_parameters())
        for n, p in enc.named_parameters():
            ref = t(g[f"grad_{tag}::{n}"])
            got = p.grad.cpu()
            # key.bias has an exactly-zero gradient (softmax is shift invariant): floor the scale at
            # 1 % of the largest gradient in the model so that bf16 noise around 0 is not "relative error"
            scale = max(1e-2 * gmax, float(ref.abs().max()))
            err = float((got - ref).abs().max()) / scale
            worst.append((err, n))
            # bf16 activations and bf16 dY/X operands: 4 % of the tensor's largest gradient
            assert err <= 4e-2, (tag, n, err, scale)
        c = torch.nn.functional.cosine_similarity(
            torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double(),
            torch.cat([t(g[f"grad_{tag}::{n}"]).reshape(-1) for n, _ in enc.named_parameters()]).double(), dim=0)
        assert float(c) >= 1 - 2e-3, (tag, float(c), sorted(worst)[-3:])


def test_encoder_gradient_accumulates_and_zero_grad(gpu):
    enc, _, _ = make_encoder(TINY, 3, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(4, 16, 1, 512, [16, 9, 3, 16])
    ids, mask = ids.to(gpu), mask.to(gpu)
    enc(input_ids=ids, attention_mask=mask).sum().backward()
    g1 = enc.flat_grads().clone()
    enc(input_ids=ids, attention_mask=mask).sum().backward()
    assert torch.allclose(enc.flat_grads(), 2 * g1, rtol=1e-3, atol=1e-5)
    enc.zero_grad()
    enc(input_ids=ids, attention_mask=mask).sum().backward()
    assert torch.allclose(enc.flat_grads(), g1, rtol=1e-3, atol=1e-5)


def test_bert_base_full_depth_vs_oracle(gpu):
    """12-layer bert-base-shaped encoder (small vocab), B=8, S=256 ragged: HIP bf16 vs CPU-oracle fp32."""
    shape_kw = dict(vocab_size=2048, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, max_position_embeddings=512)
    enc, P, cfg_o = make_encoder(shape_kw, 77, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(8, 256, 78, 2048, [256, 130, 27, 256, 77, 200, 1, 64])
    with torch.no_grad():
        ref = O.encoder_forward(P, ids, mask, cfg_o, "average", False)
        emb = enc(input_ids=ids.to(gpu), attention_mask=mask.to(gpu)).cpu()
    c = cos_rows(emb, ref)
    assert float(c.min()) >= 1 - 1e-3, c


@pytest.mark.parametrize("B,S,lengths", [(3, 27, [27, 13, 1]), (1, 1, [1]), (2, 64, [64, 2])])
def test_encoder_odd_shapes_vs_oracle(gpu, B, S, lengths):
    """sequence lengths that are not multiples of any tile size (the reference pads q and k to the batch maximum,
    27..130 tokens with the README flags), a single token, and a single sequence"""
    enc, P, cfg_o = make_encoder(TINY, 9, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(B, S, 10, 512, lengths)
    for norm in (False, True):
        Pg = {n: v.clone().requires_grad_(True) for n, v in P.items()}
        ref = O.encoder_forward(Pg, ids, mask, cfg_o, "average", norm)
        emb = enc(input_ids=ids.to(gpu), attention_mask=mask.to(gpu), normalize=norm)
        assert float(cos_rows(emb.detach().cpu(), ref.detach()).min()) >= 1 - 1e-3
        cot = O.det_uniform((B, cfg_o.hidden_size), 12)
        (ref * cot).sum().backward()
        enc.zero_grad()
        (emb * cot.to(gpu)).sum().backward()
        got = torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double()
        want = torch.cat([Pg[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
        assert float(torch.nn.functional.cosine_similarity(got, want, dim=0)) >= 1 - 3e-3


@pytest.mark.parametrize("shape_kw", [
    dict(vocab_size=1024, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096, max_position_embeddings=128),
    dict(vocab_size=1024, hidden_size=320, num_hidden_layers=2, num_attention_heads=5, intermediate_size=1280, max_position_embeddings=128),
])
def test_encoder_other_widths_vs_oracle(gpu, shape_kw):
    """bert-large width (1024: four float4 chunks per lane in the row kernels) and a width that is not a multiple of the
    256-wide GEMM tiles (320 = 5 heads; intermediate 1280): forward and every gradient against the oracle."""
    enc, P, cfg_o = make_encoder(shape_kw, 51, "average", gpu)
    enc.eval()
    B, S = 6, 96
    ids, mask = ragged_batch(B, S, 52, 1024, [96, 40, 7, 96, 65, 1])
    Pg = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ref = O.encoder_forward(Pg, ids, mask, cfg_o, "average", True)
    emb = enc(input_ids=ids.to(gpu), attention_mask=mask.to(gpu), normalize=True)
    assert float(cos_rows(emb.detach().cpu(), ref.detach()).min()) >= 1 - 1e-3
    cot = O.det_uniform((B, cfg_o.hidden_size), 53)
    (ref * cot).sum().backward()
    (emb * cot.to(gpu)).sum().backward()
    got = torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double()
    want = torch.cat([Pg[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
    assert float(torch.nn.functional.cosine_similarity(got, want, dim=0)) >= 1 - 3e-3


@pytest.mark.parametrize("case", range(8))
def test_encoder_random_shapes_vs_oracle(gpu, case):
    """Seeded random encoder shapes between the pinned ones: width (a multiple of 64 up to 1024: the row kernels' chunk
    counts and the GEMMs' ragged tiles), depth, intermediate size, batch, sequence length with ragged padding, pooling,
    normalisation -- forward and all gradients against the oracle."""
    import random
    r = random.Random(9000 + case)
    heads = r.choice([1, 2, 3, 5, 6, 9, 12, 13, 16])
    H = 64 * heads
    shape_kw = dict(vocab_size=r.choice([64, 512, 1000]), hidden_size=H, num_hidden_layers=r.choice([1, 2, 3]),
                    num_attention_heads=heads, intermediate_size=64 * r.randrange(1, 24), max_position_embeddings=320)
    pooling, norm = r.choice(["average", "cls"]), r.choice([False, True])
    B, S = r.choice([1, 2, 5, 9]), r.choice([1, 7, 31, 32, 33, 64, 100, 129, 257, 300])
    lengths = [S] + [r.randrange(1, S + 1) for _ in range(B - 1)]
    enc, P, cfg_o = make_encoder(shape_kw, 9100 + case, pooling, gpu)
    enc.eval()
    ids, mask = ragged_batch(B, S, 9200 + case, shape_kw["vocab_size"], lengths)
    Pg = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ref = O.encoder_forward(Pg, ids, mask, cfg_o, pooling, norm)
    emb = enc(input_ids=ids.to(gpu), attention_mask=mask.to(gpu), normalize=norm)
    what = (shape_kw, pooling, norm, B, S, lengths)
    assert float(cos_rows(emb.detach().cpu(), ref.detach()).min()) >= 1 - 1e-3, what
    cot = O.det_uniform((B, H), 9300 + case)
    (ref * cot).sum().backward()
    (emb * cot.to(gpu)).sum().backward()
    got = torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).cpu().double()
    want = torch.cat([Pg[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
    assert float(torch.nn.functional.cosine_similarity(got, want, dim=0)) >= 1 - 3e-3, what


def _moco_opt(model_dir, **kw):
    d = dict(queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.0, norm_doc=False, norm_query=False,
             moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False,
             retriever_model_id=model_dir, pooling="average", random_init=True, projection_size=128,
             optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, warmup_steps=2, total_steps=10,
             lr_min_ratio=0.0, scheduler="linear")
    d.update(kw)
    return types.SimpleNamespace(**d)


def _write_cfg(tmp_path, shape_kw, **extra):
    d = dict(shape_kw)
    d.update(extra)
    os.makedirs(tmp_path, exist_ok=True)
    with open(os.path.join(tmp_path, "config.json"), "w") as f:
        json.dump(d, f)
    return str(tmp_path)


@pytest.mark.parametrize("name,norm,ls,ltol,wtol", [("g8_moco_trace.npz", False, 0.0, 3e-2, 4e-3),
                                                   ("g9_moco_trace_norm.npz", True, 0.1, 1e-2, 2.5e-3)])
def test_moco_micro_steps_vs_reference_trace(golden_dir, gpu, tmp_path, name, norm, ls, ltol, wtol):
    """5 full micro-steps (q fwd/bwd, EMA, k fwd, InfoNCE, enqueue, AdamW + warm-up schedule), dropout off,
    against the traces the reference's own pieces produced: g8 = README flags (un-normalised embeddings,
    T = 0.05: step 0 has loss exactly 0 and zero gradient, later losses are 30-45), g9 = normalised
    embeddings + label smoothing 0.1 (losses 1.8-4.4, every step has a real gradient)."""
    from src import moco, utils
    g = load(golden_dir, name)
    mdir = _write_cfg(tmp_path / "tiny", TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    opt = _moco_opt(mdir, norm_query=norm, norm_doc=norm, label_smoothing=ls)
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**TINY)
    P = O.make_params(cfg_o, 71)
    model.encoder_q.load_state_dict(P)
    model.encoder_k.load_state_dict(P)
    with torch.no_grad():
        model.queue.copy_(torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0))
    model = model.cuda()
    model.train()
    optimizer, scheduler = utils.set_optim(opt, model)
    B, S = 8, 24
    for step in range(5):
        q_tokens, q_mask = ragged_batch(B, S, 7300 + step, 512, [S, 17, 3, 24, 9, 24, 12, 20])
        k_tokens, k_mask = ragged_batch(B, S, 7400 + step, 512, [20, S, 5, 11, 24, 7, 24, 16])
        loss, stats = model(q_tokens.to(gpu), q_mask.to(gpu), k_tokens.to(gpu), k_mask.to(gpu), stats_prefix="train")
        loss.backward()
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        ref_loss = float(g["losses"][step])
        # logits are O(10) with T = 0.05: bf16 q / queue / encoder give ~1e-2 absolute on the loss
        assert abs(float(loss) - ref_loss) < ltol * max(1.0, abs(ref_loss)), (step, float(loss), ref_loss)
        assert float(stats["train/accuracy"][0]) == float(g["accs"][step])
        assert int(model.queue_ptr) == int(g["ptrs"][step])
        assert abs(float(stats["train/stdq"][0]) - float(g["stdq"][step])) < 5e-3
        assert abs(float(stats["train/stdk"][0]) - float(g["stdk"][step])) < 5e-3
    qf = model.queue.cpu()
    assert float((qf - t(g["queue_final"])).abs().max()) < 2e-2
    wq = model.encoder_q.state_dict()["encoder.layer.1.output.dense.weight"].cpu()[::7, ::5]
    wk = model.encoder_k.state_dict()["encoder.layer.1.output.dense.weight"].cpu()[::7, ::5]
    # AdamW's first steps move every weight by ~lr regardless of gradient size; sign flips of tiny bf16
    # gradients show up as 2*lr differences
    assert float((wq - t(g["wq_probe"])).abs().max()) < wtol
    # ... and the update direction agrees: correlation of (w_final - w_init) with the reference's
    w0 = P["encoder.layer.1.output.dense.weight"][::7, ::5]
    dref, dgot = (t(g["wq_probe"]) - w0).double().reshape(-1), (wq - w0).double().reshape(-1)
    assert float(torch.nn.functional.cosine_similarity(dref, dgot, dim=0)) > 0.9
    assert float((wk - t(g["wk_probe"])).abs().max()) < 1e-4


@pytest.mark.parametrize("tag,ltol", [("plain", 2e-2), ("norm", 1e-2)])
def test_inbatch_vs_reference(golden_dir, gpu, tmp_path, tag, ltol):
    """`InBatch` (one encoder embeds queries AND keys, both differentiable; positives on the diagonal)
    against what the reference's src/inbatch.py produced on the same weights and batches (g10): loss, the four
    stats, and the gradient of every encoder parameter after the two backward passes have accumulated."""
    from src import inbatch
    g = load(golden_dir, "g10_inbatch.npz")
    norm, temp, ls = (float(x) for x in g[f"{tag}_cfg"])
    mdir = _write_cfg(tmp_path / "tiny", TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    opt = _moco_opt(mdir, norm_query=bool(norm), norm_doc=bool(norm), label_smoothing=ls, temperature=temp)
    model = inbatch.InBatch(opt)
    P = O.make_params(O.BertShape(**TINY), int(g["seed"]))
    model.encoder.load_state_dict(P)
    model = model.cuda()
    model.train()
    loss, stats = model(t(g["q_ids"]).to(gpu), t(g["q_mask"]).to(gpu), t(g["k_ids"]).to(gpu), t(g["k_mask"]).to(gpu),
                        stats_prefix="train", iter_stats={})
    loss.backward()
    ref = g[f"{tag}_stats"]
    assert abs(float(loss) - float(ref[0])) < ltol * max(1.0, abs(float(ref[0]))), (float(loss), float(ref[0]))
    assert float(stats["train/accuracy"][0]) == float(ref[1])
    assert abs(float(stats["train/stdq"][0]) - float(ref[2])) < 5e-3
    assert abs(float(stats["train/stdk"][0]) - float(ref[3])) < 5e-3
    gmax = max(float(t(g[f"{tag}_grad::{n}"]).abs().max()) for n in P)
    num = den_a = den_b = 0.0
    for n, p in model.encoder.named_parameters():
        r = t(g[f"{tag}_grad::{n}"]).double()
        got = p.grad.detach().cpu().double()
        scale = max(float(r.abs().max()), 0.01 * gmax)
        assert float((got - r).abs().max()) <= 0.05 * scale, (tag, n, float((got - r).abs().max()), scale)
        num += float((got * r).sum()); den_a += float((got * got).sum()); den_b += float((r * r).sum())
    assert num / (den_a ** 0.5 * den_b ** 0.5) > 0.998
    # a second step reuses both activation buffers (no growth) and accumulates on top
    n_bufs = sum(len(v) for v in model.encoder._ws_pool.values())
    loss2, _ = model(t(g["q_ids"]).to(gpu), t(g["q_mask"]).to(gpu), t(g["k_ids"]).to(gpu), t(g["k_mask"]).to(gpu))
    loss2.backward()
    assert sum(len(v) for v in model.encoder._ws_pool.values()) == n_bufs
    w = model.encoder.get_parameter("encoder.layer.1.output.dense.weight")
    assert torch.allclose(w.grad.cpu(), 2 * t(g[f"{tag}_grad::encoder.layer.1.output.dense.weight"]), atol=0.1 * gmax)


def test_moco_checkpoint_layout_roundtrip(gpu, tmp_path):
    from src import moco, utils
    mdir = _write_cfg(tmp_path / "tiny", TINY)
    opt = _moco_opt(mdir, output_dir=str(tmp_path / "out"))
    model = moco.MoCo(opt).cuda()
    optimizer, scheduler = utils.set_optim(opt, model)
    ids, mask = ragged_batch(8, 16, 1, 512)
    loss, _ = model(ids.to(gpu), mask.to(gpu), ids.to(gpu), mask.to(gpu))
    loss.backward()
    optimizer.step(); scheduler.step()
    utils.save(model, optimizer, scheduler, 1, opt, str(tmp_path / "out"), "step-1")
    latest = tmp_path / "out" / "checkpoint" / "latest"
    assert os.path.islink(latest)
    ck = torch.load(latest / "checkpoint.pth", map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == ["model", "opt", "optimizer", "scheduler", "step"]
    keys = list(ck["model"].keys())
    n_enc = len(O.param_names(O.BertShape(**TINY)))
    assert keys[:2] == ["queue", "queue_ptr"] and len(keys) == 2 + 2 * n_enc
    assert tuple(ck["model"]["queue"].shape) == (128, 64) and ck["model"]["queue_ptr"].dtype == torch.int64
    assert all(k.startswith(("encoder_q.", "encoder_k.")) for k in keys[2:])
    model2, opt2, sch2, opt_ck, step = utils.load(moco.MoCo, str(latest), opt)
    assert step == 1
    for (n1, a), (n2, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert n1 == n2 and torch.equal(a.cpu(), b.cpu()), n1
    sd1, sd2 = optimizer.state_dict(), opt2.state_dict()
    assert sd1["param_groups"] == sd2["param_groups"]
    for k in sd1["state"]:
        assert torch.equal(sd1["state"][k]["exp_avg"].cpu(), sd2["state"][k]["exp_avg"].cpu())


def test_base_width_backward_vs_reference_golden(golden_dir, gpu):
    """bert-base WIDTH (H 768, 12 heads, I 3072, S 256 ragged) backward against the REFERENCE's gradients (g12: a subsample
    of every tensor + its L2 norm; made by oracle/make_golden.py from src.contriever.Contriever on HF BertModel)."""
    g = load(golden_dir, "g12_base_grads.npz")
    ids, mask, cot = t(g["ids"]).to(gpu), t(g["mask"]).to(gpu), t(g["cot"]).to(gpu)
    for pooling, norm in (("average", False), ("average", True)):
        enc, _, cfg_o = make_encoder(BASE1, int(g["seed"]), pooling, gpu)
        enc.eval()
        emb = enc(input_ids=ids, attention_mask=mask, normalize=norm)
        tag = f"{pooling}_{int(norm)}"
        assert float(cos_rows(emb.detach().cpu(), t(g[f"emb_{tag}"])).min()) >= 1 - 1e-3
        (emb * cot).sum().backward()
        gmax = max(float(np.abs(g[f"grad_{tag}::{n}"]).max()) for n, _ in enc.named_parameters())
        num = da = db = 0.0
        for n, p in enc.named_parameters():
            stat = g[f"gstat_{tag}::{n}"]
            ref = t(g[f"grad_{tag}::{n}"]).double()
            got = p.grad.detach().reshape(-1)[::int(stat[2])].cpu().double()
            scale = max(1e-2 * gmax, float(ref.abs().max()))
            assert float((got - ref).abs().max()) <= 5e-2 * scale, (tag, n, float((got - ref).abs().max()), scale)
            nrm = float(p.grad.detach().double().norm())
            assert abs(nrm - stat[1]) <= 3e-2 * stat[1] + 1e-3 * gmax, (tag, n, nrm, stat[1])
            num += float((got * ref).sum()); da += float((got * got).sum()); db += float((ref * ref).sum())
        assert num / (da ** 0.5 * db ** 0.5) >= 1 - 2e-3, (tag, num / (da ** 0.5 * db ** 0.5))


def test_bert_base_full_depth_backward_vs_oracle(gpu):
    """The shapes the bench runs (12 layers, H 768, I 3072, 12 heads, S 256) at B = 8, ragged: every parameter gradient of
    the HIP bf16 backward against the fp32 CPU oracle -- global cosine, and per-tensor cosine for the weight matrices."""
    shape_kw = dict(vocab_size=2048, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, max_position_embeddings=512)
    enc, P, cfg_o = make_encoder(shape_kw, 77, "average", gpu)
    enc.eval()
    B = 8
    ids, mask = ragged_batch(B, 256, 78, 2048, [256, 130, 27, 256, 77, 200, 1, 64])
    cot = O.det_uniform((B, 768), 79)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    Pg = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ref = O.encoder_forward(Pg, ids, mask, cfg_o, "average", False)
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


def test_weights_written_on_the_gpu_reach_the_bf16_mirror(gpu):
    """forward, then load_state_dict / an in-place parameter write on the GPU-resident encoder, then forward: the matrix
    products must use the NEW weights (the bf16 mirror is rebuilt), i.e. equal a fresh encoder built from them."""
    enc, P, cfg_o = make_encoder(TINY, 5, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(4, 16, 1, 512, [16, 9, 3, 16])
    ids, mask = ids.to(gpu), mask.to(gpu)
    with torch.no_grad():
        e0 = enc(input_ids=ids, attention_mask=mask).clone()
        P2 = O.make_params(cfg_o, 6)
        enc.load_state_dict(P2)                                    # on the GPU, after a forward has built the mirror
        e1 = enc(input_ids=ids, attention_mask=mask).clone()
        fresh, _, _ = make_encoder(TINY, 6, "average", gpu)
        fresh.eval()
        want = fresh(input_ids=ids, attention_mask=mask)
        assert torch.equal(e1, want) and not torch.equal(e0, e1)
        w = enc.get_parameter("encoder.layer.0.intermediate.dense.weight")
        w.mul_(1.5)                                                # a torch-side in-place write through a parameter view
        e2 = enc(input_ids=ids, attention_mask=mask)
        fw = fresh.get_parameter("encoder.layer.0.intermediate.dense.weight")
        fw.mul_(1.5)
        assert torch.equal(e2, fresh(input_ids=ids, attention_mask=mask)) and not torch.equal(e2, e1)


@pytest.mark.parametrize("shape_kw,B,S,lengths", [(TINY, 8, 24, [24, 17, 3, 24, 9, 24, 12, 20]),
                                                   (BASE1, 6, 256, [256, 130, 27, 256, 77, 200])])
def test_key_encoder_split_last_layer_is_bitwise_neutral(gpu, shape_kw, B, S, lengths):
    """czc_encoder_forward(split_batch): the last layer + pooling in two batch ranges (so that the first half of the keys
    can be all-gathered while the second half is still being encoded) must give bit-identical embeddings, and the event
    recorded behind the first range must fire."""
    import ctypes
    from czech_contriever_amd import lib as L
    enc, _, _ = make_encoder(shape_kw, 7, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(B, S, 8, shape_kw["vocab_size"], lengths)
    ids, mask = ids.to(gpu), mask.to(gpu)
    ev = ctypes.c_void_p()
    L.check(L.load().czc_event_create(ctypes.byref(ev)))
    with torch.no_grad():
        whole = enc.encode(ids, mask, normalize=True).clone()
        for n in (B // 2, 1, B - 1):
            if (n * S) % 16:
                continue
            side = torch.cuda.Stream()
            part = enc.encode(ids, mask, normalize=True, split=(n, ev))
            L.check(L.load().czc_stream_wait_event(side.cuda_stream, ev))     # a side stream may read rows [0, n) from here on
            with torch.cuda.stream(side):
                first = part[:n].clone()
            side.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(part, whole), (n, float((part - whole).abs().max()))
            assert torch.equal(first, whole[:n])
    L.check(L.load().czc_event_destroy(ev))
    with pytest.raises(RuntimeError):                              # the split is for passes nothing differentiates through
        enc.train()
        enc._forward_impl(ids, mask, False, save=True, split=None)
        L.check(L.load().czc_encoder_forward(ctypes.byref(enc._shape), enc._flat.data_ptr(), enc.mirror().data_ptr(), ids.data_ptr(),
                                             mask.data_ptr(), B, S, 0, 0, 0, 0, 1, enc._workspace(B, S, True).data_ptr(),
                                             enc._workspace(B, S, True).numel(), whole.data_ptr(), B // 2, None, L.stream_ptr()))


@pytest.mark.parametrize("B,S,lengths", [(3, 27, [27, 13, 1]), (2, 200, [200, 77]), (5, 64, [64, 2, 33, 64, 17])])
def test_dropout_backward_never_reads_unwritten_workspace(gpu, B, S, lengths):
    """the forward of a saved pass hands the backward its attention-dropout keep decisions as bit words in the
    activation workspace, and skips whole query blocks beyond a sequence's padded length (their words stay unwritten).
    The backward must never read what the forward did not write: with the workspace pre-filled with 0xFF bytes and with
    0x00 bytes, at odd lengths below 256 and dropout on, embeddings and every gradient must come out bit-identical
    (weight matrices) / equal to the atomics' rounding (vectors and embeddings)."""
    shape_kw = dict(TINY, max_position_embeddings=256)
    from src.contriever import Contriever, bert_config
    cfg = bert_config(vocab_size=shape_kw["vocab_size"], hidden_size=shape_kw["hidden_size"],
                      num_hidden_layers=shape_kw["num_hidden_layers"], num_attention_heads=shape_kw["num_attention_heads"],
                      intermediate_size=shape_kw["intermediate_size"], max_position_embeddings=256,
                      hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.25)
    enc = Contriever(cfg, pooling="average")
    enc.load_state_dict(O.make_params(O.BertShape(**shape_kw), 91), strict=True)
    enc = enc.cuda().train()
    ids, mask = ragged_batch(B, S, 9100, shape_kw["vocab_size"], lengths)
    ids, mask = ids.to(gpu), mask.to(gpu)
    w = O.det_uniform((B, shape_kw["hidden_size"]), 92).to(gpu)
    outs = []
    for fill in (0xFF, 0x00):
        enc.zero_grad(set_to_none=True)
        enc._ws_pool.clear()
        enc._workspace(B, S, True).fill_(fill)          # the buffer the next differentiable forward leases
        enc._drop_calls = 0                             # the same dropout stream for both passes
        emb = enc(input_ids=ids, attention_mask=mask, normalize=True)
        (emb * w).sum().backward()
        torch.cuda.synchronize()
        outs.append((emb.detach().clone(), enc.flat_grads().clone()))
    (e0, g0), (e1, g1) = outs
    assert torch.isfinite(g0).all() and torch.isfinite(g1).all()
    assert torch.equal(e0, e1)
    nm = enc._n_mat
    assert torch.equal(g0[:nm], g1[:nm])                                   # split-K weight gradients: fixed summation order
    scale = float(g0[nm:].abs().max()) + 1e-12
    assert float((g0[nm:] - g1[nm:]).abs().max()) <= 1e-5 * scale          # fp32 atomics: order-dependent last bits only


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_key_encoder_on_second_stream_changes_nothing(gpu, tmp_path, dropout):
    """MoCo.key_overlap: EMA + key-encoder forward on a second HIP stream beside the query encoder's forward (one GPU).
    Same kernels on the same inputs, only their interleaving differs: loss trace, queue, pointer and key-encoder weights
    bit-identical at step 0, everything equal to the float atomics' rounding afterwards."""
    from src import moco, utils
    mdir = _write_cfg(tmp_path / "tiny", TINY, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    res = []
    for overlap in (False, True):
        torch.manual_seed(0)
        opt = _moco_opt(mdir, norm_query=True, norm_doc=True)
        model = moco.MoCo(opt)
        P = O.make_params(O.BertShape(**TINY), 71)
        model.encoder_q.load_state_dict(P)
        model.encoder_k.load_state_dict(P)
        model = model.cuda().train()
        model.key_overlap = overlap
        optimizer, scheduler = utils.set_optim(opt, model)
        losses, queue0 = [], None
        for step in range(4):
            qt, qm = ragged_batch(8, 24, 7300 + step, 512, [24, 17, 3, 24, 9, 24, 12, 20])
            kt, km = ragged_batch(8, 24, 7400 + step, 512, [20, 24, 5, 11, 24, 7, 24, 16])
            loss, _ = model(qt.to(gpu), qm.to(gpu), kt.to(gpu), km.to(gpu), stats_prefix="train")
            loss.backward()
            optimizer.step(); scheduler.step(); model.zero_grad()
            losses.append(float(loss.detach()))
            if step == 0:
                queue0 = model.queue.clone()
        torch.cuda.synchronize()
        assert (model._key_stream is not None) == overlap
        res.append((losses, queue0.cpu(), model.queue.cpu(), int(model.queue_ptr), model.encoder_q.flat_params.cpu(),
                    model.encoder_k.flat_params.cpu()))
    (l0, q00, qa, pa, wqa, wka), (l1, q01, qb, pb, wqb, wkb) = res
    assert l0[0] == l1[0] and torch.equal(q00, q01) and pa == pb == 32
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 2e-3
    assert float((qa - qb).abs().max()) < 1e-3
    for a, b in ((wqa, wqb), (wka, wkb)):
        d = (a - b).abs()
        assert float(d.max()) < 2.5e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))


def test_no_grad_pass_and_differentiable_pass_agree(gpu):
    """One encoder, dropout 0, same batch: the no_grad pass (key encoder / inference: FFN-up keeps gelu(u) only, evaluated on
    the fp32 sum -- CZC_EPI_BIAS_GELU_ONLY) against the differentiable pass (gelu and gelu' evaluated on the bf16-rounded
    pre-activation, which the backward needs).  The two roundings differ inside bf16 noise; nothing of the reference pins
    the asymmetry (it runs one fp32 path for both), so it is bounded here: cosine >= 1 - 1e-5 per row, max |diff| <= 1 % of
    the embedding scale -- two orders inside the 1e-3 bar both passes meet against the fp32 oracle."""
    enc, P, cfg_o = make_encoder(BASE1, 22, "average", gpu)
    enc.eval()
    ids, mask = ragged_batch(4, 256, 23, 1024, [256, 130, 256, 77])
    ids, mask = ids.to(gpu), mask.to(gpu)
    with torch.no_grad():
        a = enc(input_ids=ids, attention_mask=mask).cpu()
    b = enc(input_ids=ids, attention_mask=mask)
    assert b.requires_grad and not a.requires_grad
    b = b.detach().cpu()
    c = cos_rows(a, b)
    assert float(c.min()) >= 1 - 1e-5, c
    assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max())
    ref = O.encoder_forward(P, ids.cpu(), mask.cpu(), cfg_o, "average", False)
    assert float(cos_rows(a, ref).min()) >= 1 - 1e-3 and float(cos_rows(b, ref).min()) >= 1 - 1e-3
