"""The command-line path on the GPU: train.py (config-1-sized plumbing run on synthetic ids), checkpoints,
auto-resume, and the weight_decay_from_init penalty against the oracle."""
import json
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moco_bert_oracle as O
from oracle.make_golden import TINY, ragged_batch


def _cfg_dir(tmp_path, **extra):
    d = str(tmp_path / "model")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(TINY, **extra), f)
    return d


def test_train_cli_runs_saves_and_resumes(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import train
    mdir = _cfg_dir(tmp_path)
    out = str(tmp_path / "run")
    args = ["--retriever_model_id", mdir, "--random_init", "--output_dir", out, "--save_dir", out,
            "--chunk_length", "32", "--per_gpu_batch_size", "8", "--target_batch_size", "16", "--queue_size", "64",
            "--projection_size", "128", "--temperature", "0.05", "--momentum", "0.99", "--lr", "1e-3",
            "--warmup_steps", "2", "--total_steps", "4", "--save_freq", "2", "--log_freq", "1", "--norm_query", "--norm_doc"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    train.main(args)
    ck = os.path.join(out, "checkpoint")
    assert os.path.islink(os.path.join(ck, "latest"))
    assert os.path.exists(os.path.join(ck, "step-2", "checkpoint.pth")) and os.path.exists(os.path.join(ck, "step-4", "checkpoint.pth"))
    assert os.path.exists(os.path.join(out, "opt.txt"))
    c4 = torch.load(os.path.join(ck, "step-4", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert c4["step"] == 4 and c4["opt"].queue_size == 64
    # update_freq = 16 / 8 = 2 micro-steps per optimizer step: 4 steps enqueued 8 micro-batches of 8 keys
    assert int(c4["model"]["queue_ptr"]) == (8 * 8) % 64
    assert c4["optimizer"]["state"][0]["step"] == 4
    # auto-resume: the output directory exists -> continues from `latest` up to the new total_steps
    args2 = [a if a != "4" else "6" for a in args]
    seen = {}
    loop = train.train

    def spy(opt, model, optimizer, scheduler, step):
        m = model.module if hasattr(model, "module") else model
        seen.update(q=m.encoder_q.dropout_state(), k=m.encoder_k.dropout_state(), step=step)
        return loop(opt, model, optimizer, scheduler, step)
    monkeypatch.setattr(train, "train", spy)
    train.main(args2)
    # the resumed run continues the dropout streams where step 4 stood (8 micro-steps) instead of replaying steps 0..3
    assert seen == {"q": {"salt": 0, "calls": 8}, "k": {"salt": 1, "calls": 8}, "step": 4}
    c6 = torch.load(os.path.join(ck, "step-6", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert c6["step"] == 6
    # like the reference, a resumed run keeps the CHECKPOINT's schedule (utils.load builds the optimizer from
    # opt_checkpoint, src/utils.py:86-88): total_steps was 4 there, so lambda(4..5) = 0 and the weights stand still,
    # while the Adam moments, the EMA and the queue keep moving
    w4 = c4["model"]["encoder_q.encoder.layer.0.output.dense.weight"]
    w6 = c6["model"]["encoder_q.encoder.layer.0.output.dense.weight"]
    assert torch.equal(w4, w6)
    assert not torch.equal(c4["optimizer"]["state"][5]["exp_avg"], c6["optimizer"]["state"][5]["exp_avg"])
    assert not torch.equal(c4["model"]["queue"], c6["model"]["queue"])
    # the reference's loader strips `encoder_q.` and loads a plain Contriever (src/contriever.py:126-144)
    from src.contriever import load_retriever
    enc, tok, mid = load_retriever(os.path.join(ck, "step-6"))
    assert torch.equal(enc.state_dict()["encoder.layer.0.output.dense.weight"], w6)


def test_resumed_run_follows_the_uninterrupted_trajectory(tmp_path):
    """Dropout ON: six optimizer steps in one go against three steps, utils.save, utils.load into fresh objects,
    set_dropout_position and three more.  Everything a step depends on must survive the checkpoint: both encoders, the
    queue and its pointer, the Adam moments and step count, the schedule, and the position of the two dropout streams.
    The two runs may differ by the float atomics' last bits only (embedding / LayerNorm / bias gradients: DESIGN.md section
    4 (9)), which two uninterrupted runs differ by as well -- measured here as the yardstick; a replayed or shifted dropout
    stream, a lost moment or a reset schedule moves most weights by ~lr per step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src import moco, utils
    mdir = _cfg_dir(tmp_path)                      # TINY with its default dropout 0.1 / 0.1
    dev = torch.device("cuda:0")

    def opts():
        return types.SimpleNamespace(
            queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.1, norm_doc=True, norm_query=True,
            moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
            pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98,
            eps=1e-6, warmup_steps=2, total_steps=8, lr_min_ratio=0.0, scheduler="linear", seed=0)

    def batch(i):
        qt, qm = ragged_batch(8, 24, 8300 + i, 512, [24, 17, 3, 24, 9, 24, 12, 20])
        kt, km = ragged_batch(8, 24, 8400 + i, 512, [20, 24, 5, 11, 24, 7, 24, 16])
        return qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev)

    def fresh():
        torch.manual_seed(0)
        opt = opts()
        model = moco.MoCo(opt)
        P = O.make_params(O.BertShape(**TINY), 71)
        model.encoder_q.load_state_dict(P)
        model.encoder_k.load_state_dict(P)
        with torch.no_grad():
            model.queue.copy_(torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0))
        model = model.to(dev).train()
        optimizer, scheduler = utils.set_optim(opt, model)
        return opt, model, optimizer, scheduler

    def steps(model, optimizer, scheduler, lo, hi):
        for i in range(lo, hi):
            loss, _ = model(*batch(i), stats_prefix="train")
            loss.backward()
            optimizer.step()
            scheduler.step()
            model.zero_grad()

    def state(model, optimizer):
        torch.cuda.synchronize()
        sd = optimizer.state_dict()["state"]
        return (model.encoder_q.flat_params.cpu().clone(), model.encoder_k.flat_params.cpu().clone(), model.queue.cpu().clone(),
                int(model.queue_ptr), torch.cat([sd[k]["exp_avg"].reshape(-1).cpu() for k in sorted(sd)]))

    runs = []
    for _ in range(2):                              # two uninterrupted runs: the yardstick
        opt, model, optimizer, scheduler = fresh()
        steps(model, optimizer, scheduler, 0, 6)
        runs.append(state(model, optimizer))
    opt, model, optimizer, scheduler = fresh()
    steps(model, optimizer, scheduler, 0, 3)
    utils.save(model, optimizer, scheduler, 3, opt, str(tmp_path), "step-3")
    del model, optimizer, scheduler
    torch.manual_seed(0)                            # (train.py seeds before it builds or loads anything)
    model, optimizer, scheduler, opt2, step = utils.load(moco.MoCo, os.path.join(str(tmp_path), "checkpoint", "step-3"), opts())
    assert step == 3
    model.train()
    model.set_dropout_position(step * 1)            # train.py: step * update_freq micro-steps
    steps(model, optimizer, scheduler, 3, 6)
    resumed = state(model, optimizer)

    assert resumed[3] == runs[0][3] == runs[1][3]
    for idx, name in ((0, "query weights"), (1, "key weights"), (2, "queue"), (4, "first moments")):
        noise = (runs[0][idx] - runs[1][idx]).abs()
        d = (resumed[idx] - runs[0][idx]).abs()
        scale = float(runs[0][idx].abs().max())
        # within a few times what two uninterrupted runs differ by (floor: 1e-6 of the tensor's scale, for runs that agree bit for bit)
        assert float(d.mean()) <= 4.0 * float(noise.mean()) + 1e-6 * scale, (name, float(d.mean()), float(noise.mean()))
        assert float(d.max()) <= 4.0 * float(noise.max()) + 2e-3 * scale, (name, float(d.max()), float(noise.max()))


def test_train_cli_inbatch_mode(tmp_path, monkeypatch):
    """--contrastive_mode inbatch (reference train.py:451-452) through the same loop: checkpoint keys are `encoder.*`."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import train
    mdir = _cfg_dir(tmp_path)
    out = str(tmp_path / "run_inbatch")
    args = ["--retriever_model_id", mdir, "--random_init", "--output_dir", out, "--save_dir", out,
            "--contrastive_mode", "inbatch", "--chunk_length", "32", "--per_gpu_batch_size", "8",
            "--target_batch_size", "8", "--temperature", "0.05", "--lr", "1e-3", "--warmup_steps", "1",
            "--total_steps", "3", "--save_freq", "3", "--log_freq", "1", "--norm_query", "--norm_doc"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    train.main(args)
    ck = torch.load(os.path.join(out, "checkpoint", "step-3", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert ck["step"] == 3
    keys = list(ck["model"].keys())
    assert all(k.startswith("encoder.") for k in keys) and "encoder.encoder.layer.1.output.dense.weight" in keys
    assert ck["optimizer"]["state"][0]["step"] == 3


def test_train_cli_on_jsonl_corpus(tmp_path, monkeypatch):
    """--train_data: jsonl -> local WordPiece vocab -> src.data (Dataset, random crops, Collator) -> DataLoader -> the
    GPU step, with queries and keys padded to different widths."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import random
    import train
    mdir = _cfg_dir(tmp_path)
    words = [f"w{i}" for i in range(507)]
    with open(os.path.join(mdir, "vocab.txt"), "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words) + "\n")
    rng = random.Random(0)
    corpus = tmp_path / "corpus.jsonl"
    with open(corpus, "w") as f:
        for _ in range(40):
            f.write(json.dumps({"text": " ".join(rng.choice(words) for _ in range(rng.randint(12, 60)))}) + "\n")
    out = str(tmp_path / "run_corpus")
    args = ["--retriever_model_id", mdir, "--random_init", "--output_dir", out, "--save_dir", out, "--train_data", str(corpus),
            "--chunk_length", "32", "--per_gpu_batch_size", "8", "--target_batch_size", "8", "--queue_size", "64",
            "--projection_size", "128", "--temperature", "0.05", "--momentum", "0.99", "--lr", "1e-3", "--warmup_steps", "1",
            "--total_steps", "7", "--save_freq", "7", "--log_freq", "1", "--norm_query", "--norm_doc", "--num_workers", "0"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    train.main(args)   # 40 documents / batch 8 = 5 batches per epoch: 7 steps cross an epoch boundary
    ck = torch.load(os.path.join(out, "checkpoint", "step-7", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert ck["step"] == 7 and int(ck["model"]["queue_ptr"]) == (7 * 8) % 64
    assert torch.isfinite(ck["model"]["encoder_q.encoder.layer.0.output.dense.weight"]).all()


def test_weight_decay_from_init_matches_oracle(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src import moco, utils
    dev = torch.device("cuda:0")
    mdir = _cfg_dir(tmp_path, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    wd = 0.5   # large on purpose: the penalty must move the loss visibly after one step
    opt = types.SimpleNamespace(
        queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.0, norm_doc=True, norm_query=True,
        moco_train_mode_encoder_k=False, weight_decay=wd, weight_decay_from_init=True, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-2, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=0, total_steps=10, lr_min_ratio=1.0, scheduler="linear")
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**TINY)
    P = O.make_params(cfg_o, 71)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    model.init_weights = {k: v.clone() for k, v in model.encoder_q.state_dict().items()}
    queue = torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.to(dev).train()
    # weight_decay_from_init -> AdamW built WITHOUT a weight_decay argument (reference src/utils.py:139-142): torch's
    # default decoupled decay 0.01 stays on
    optimizer, scheduler = utils.set_optim(opt, model)
    assert optimizer.param_groups[0]["weight_decay"] == 0.01
    st = O.MoCoState(cfg_o, P, queue, momentum=0.99, temperature=0.05, norm_query=True, norm_doc=True, l2_from_init=wd)
    for step in range(3):
        qt, qm = ragged_batch(8, 24, 7300 + step, 512, [24, 17, 3, 24, 9, 24, 12, 20])
        kt, km = ragged_batch(8, 24, 7400 + step, 512, [20, 24, 5, 11, 24, 7, 24, 16])
        loss, _ = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev))
        loss.backward()
        optimizer.step(); scheduler.step(); model.zero_grad()
        ref, _, _, _ = st.forward_backward(qt, qm, kt, km)
        st.optimizer_step(1e-2, weight_decay=0.01)
        assert abs(float(loss.detach()) - float(ref)) < 2e-2 * max(1.0, abs(float(ref))), (step, float(loss.detach()), float(ref))
    assert float(loss.detach()) > 1.0   # the penalty term is active (InfoNCE alone is < 1 here at step 2)


BERT_BASE_UNCASED = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                         intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2)


def test_config1_sized_run_bert_base_uncased_on_jsonl(tmp_path, monkeypatch):
    """BASELINE.json configs[0] at its STATED sizes: bert-base-uncased-sized encoder (12 layers, H 768, vocab 30 522),
    chunk_length 64, per_gpu_batch_size 4, queue_size 512, 100 steps on a synthetic jsonl, through the command line
    (jsonl -> WordPiece -> crops -> Collator -> DataLoader -> GPU micro-step -> AdamW -> checkpoint), dropout 0.1 on as
    the reference trains.  Weights are random-init (no checkpoints offline) and the vocabulary is synthetic."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import random
    import train
    mdir = str(tmp_path / "bert-base-uncased")
    os.makedirs(mdir)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(BERT_BASE_UNCASED, f)
    words = [f"w{i}" for i in range(BERT_BASE_UNCASED["vocab_size"] - 5)]
    with open(os.path.join(mdir, "vocab.txt"), "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words) + "\n")
    rng = random.Random(1)
    corpus = tmp_path / "corpus.jsonl"
    with open(corpus, "w") as f:
        for _ in range(600):                                   # >= 512 lines
            f.write(json.dumps({"text": " ".join(rng.choice(words) for _ in range(rng.randint(40, 200)))}) + "\n")
    out = str(tmp_path / "run_config1")
    args = ["--retriever_model_id", mdir, "--random_init", "--output_dir", out, "--save_dir", out, "--train_data", str(corpus),
            "--chunk_length", "64", "--per_gpu_batch_size", "4", "--target_batch_size", "4", "--queue_size", "512",
            "--temperature", "0.05", "--momentum", "0.9995", "--lr", "1e-5", "--warmup_steps", "20", "--total_steps", "100",
            "--save_freq", "100", "--log_freq", "20", "--num_workers", "0"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    train.main(args)
    ck = torch.load(os.path.join(out, "checkpoint", "step-100", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert ck["step"] == 100 and int(ck["model"]["queue_ptr"]) == (100 * 4) % 512
    assert tuple(ck["model"]["queue"].shape) == (768, 512)
    assert len(ck["model"]) == 2 + 2 * 197                     # 396 tensors at bert-base
    assert tuple(ck["model"]["encoder_q.embeddings.word_embeddings.weight"].shape) == (30522, 768)
    for n, v in ck["model"].items():
        assert torch.isfinite(v).all(), n
    # every one of the 100 x 4 keys replaced an initial queue column... 400 of the 512 columns are encoder outputs now
    q = ck["model"]["queue"]
    assert float(((q[:, :400].norm(dim=0) - 1.0).abs() > 1e-3).float().mean()) > 0.99     # un-normalised keys (README flags)
    assert float((q[:, 400:].norm(dim=0) - 1.0).abs().max()) < 1e-4                        # untouched initial columns


def test_config1_sized_steps_match_oracle(tmp_path):
    """The same sizes (bert-base-uncased shape incl. the 30 522-row vocabulary, S 64, B 4, K 512, README flags), dropout
    off: three full micro-steps (q fwd/bwd, EMA, k fwd, InfoNCE, enqueue, AdamW) against the fp32 CPU oracle
    (oracle.MoCoState): loss, accuracy, stdq / stdk, queue pointer, and the pooled q embeddings within 1e-3 cosine."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src import moco, utils
    dev = torch.device("cuda:0")
    mdir = str(tmp_path / "bert-base-uncased")
    os.makedirs(mdir)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(dict(BERT_BASE_UNCASED, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), f)
    opt = types.SimpleNamespace(
        queue_size=512, momentum=0.9995, temperature=0.05, label_smoothing=0.0, norm_doc=False, norm_query=False,
        moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=768, optim="adamw", lr=1e-5, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=0, total_steps=100, lr_min_ratio=1.0, scheduler="linear")
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**BERT_BASE_UNCASED)
    P = O.make_params(cfg_o, 91)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    queue = O.init_queue(768, 512, 0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.to(dev).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    st = O.MoCoState(cfg_o, P, queue, momentum=0.9995, temperature=0.05)
    B, S = 4, 64
    for step in range(3):
        qt, qm = ragged_batch(B, S, 9300 + step, 30522, [64, 41, 64, 17])
        kt, km = ragged_batch(B, S, 9400 + step, 30522, [52, 64, 9, 64])
        q_dev = None
        if step == 0:
            with torch.no_grad():
                q_dev = model.encoder_q(input_ids=qt.to(dev), attention_mask=qm.to(dev)).cpu()
        loss, stats = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train", iter_stats={})
        loss.backward()
        optimizer.step(); scheduler.step(); model.zero_grad()
        ref, q_ref, k_ref, rs = st.forward_backward(qt, qm, kt, km)
        st.optimizer_step(1e-5)
        if q_dev is not None:
            c = torch.nn.functional.cosine_similarity(q_dev.double(), q_ref.double(), dim=-1)
            assert float(c.min()) >= 1 - 1e-3, c
        # un-normalised embeddings at T = 0.05: logits are O(100); bf16 operands give ~1e-2 relative on the loss
        assert abs(float(loss.detach()) - float(ref)) < 3e-2 * max(1.0, abs(float(ref))), (step, float(loss.detach()), float(ref))
        assert abs(float(stats["train/accuracy"][0]) - float(rs["accuracy"])) <= 25.0 + 1e-4     # B = 4: one near-tie row
        assert abs(float(stats["train/stdq"][0]) - float(rs["stdq"])) < 1e-2 * max(1.0, float(rs["stdq"]))
        assert abs(float(stats["train/stdk"][0]) - float(rs["stdk"])) < 1e-2 * max(1.0, float(rs["stdk"]))
        assert int(model.queue_ptr) == st.ptr == ((step + 1) * B) % 512
    assert float((model.queue.cpu() - st.queue).abs().max()) < 3e-2 * float(st.queue.abs().max())


def test_headline_config_micro_steps_match_oracle(tmp_path):
    """BASELINE.json configs[1] (the bench's workload) at its stated shapes -- bert-base-sized encoder (12 layers, H 768, I 3072,
    vocabulary 30 522), chunk_length 256, queue_size 131 072, temperature 0.05, momentum 0.9995, README flags -- with the
    per-GPU batch cut from 256 to 32 so that the fp32 CPU oracle finishes three full micro-steps in about a minute:
    q fwd/bwd, EMA, k fwd, InfoNCE against the 131 072 slots, enqueue, AdamW.  Dropout off.  Step 0 is the saturated case
    (loss and gradient exactly 0 in the reference); from step 1 on the keys enqueued before are hard negatives."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src import moco, utils
    dev = torch.device("cuda:0")
    mdir = str(tmp_path / "bert-base-sized")
    os.makedirs(mdir)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(dict(BERT_BASE_UNCASED, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), f)
    K = 131072
    opt = types.SimpleNamespace(
        queue_size=K, momentum=0.9995, temperature=0.05, label_smoothing=0.0, norm_doc=False, norm_query=False,
        moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=768, optim="adamw", lr=1e-5, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=0, total_steps=100, lr_min_ratio=1.0, scheduler="linear")
    model = moco.MoCo(opt)
    cfg_o = O.BertShape(**BERT_BASE_UNCASED)
    P = O.make_params(cfg_o, 93)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    queue = O.init_queue(768, K, 0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.to(dev).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    st = O.MoCoState(cfg_o, P, queue, momentum=0.9995, temperature=0.05)
    B, S = 32, 256
    lens_q = [256, 130, 27, 256, 77, 200, 256, 64] * 4
    lens_k = [200, 256, 64, 31, 256, 256, 130, 99] * 4
    for step in range(3):
        qt, qm = ragged_batch(B, S, 9500 + step, 30522, lens_q)
        kt, km = ragged_batch(B, S, 9600 + step, 30522, lens_k)
        loss, stats = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train", iter_stats={})
        loss.backward()
        g_dev = torch.cat([p.grad.reshape(-1) for p in model.encoder_q.parameters()]).cpu().double()   # HF parameter order
        optimizer.step(); scheduler.step(); model.zero_grad()
        ref, q_ref, k_ref, rs = st.forward_backward(qt, qm, kt, km)
        g_ref = torch.cat([st.Pq[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)]).double()
        st.optimizer_step(1e-5)
        got = float(loss.detach())
        if step == 0:
            assert float(ref) == 0.0 and got == 0.0 and float(g_dev.abs().max()) == 0.0 and float(g_ref.abs().max()) == 0.0
        else:
            # the loss is a difference of logits of size ~1e4 (un-normalised embeddings, T = 0.05): bf16 operands leave a few
            # units of it; the gradient's direction is the robust quantity
            assert abs(got - float(ref)) < 5e-2 * abs(float(ref)) + 2.0, (step, got, float(ref))
            c = float(torch.nn.functional.cosine_similarity(g_dev, g_ref, dim=0))
            assert c > 0.98, (step, c)
            assert abs(float(g_dev.norm()) / float(g_ref.norm()) - 1) < 0.1
        assert abs(float(stats["train/accuracy"][0]) - float(rs["accuracy"])) <= 100.0 / B + 1e-4
        assert abs(float(stats["train/stdq"][0]) - float(rs["stdq"])) < 1e-2 * max(1.0, float(rs["stdq"]))
        assert abs(float(stats["train/stdk"][0]) - float(rs["stdk"])) < 1e-2 * max(1.0, float(rs["stdk"]))
        assert int(model.queue_ptr) == st.ptr == ((step + 1) * B) % K
    assert float((model.queue.cpu() - st.queue).abs().max()) < 3e-2 * float(st.queue.abs().max())


def test_momentum_update_folded_into_the_optimizer_pass_changes_nothing(tmp_path):
    """MoCo.allow_ema_fold(True): optimizer.step() applies the NEXT forward's momentum update (czc_adamw_ema_step) and that
    forward skips its own.  From one state (weights, moments, gradients), the two orders -- step, then the forward's EMA;
    and the folded step, then the forward's (skipped) EMA -- must leave both encoders, both bf16 mirrors and the moments bit
    for bit the same, in one launch and in the matrix / tail pair of launches DataParallel uses; a checkpoint in between is
    refused; a torch-side write to either encoder after the step cancels the skip."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src import moco, utils
    mdir = _cfg_dir(tmp_path)
    dev = torch.device("cuda:0")
    opt = types.SimpleNamespace(
        queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.1, norm_doc=True, norm_query=True,
        moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=2, total_steps=10, lr_min_ratio=0.0, scheduler="linear")
    torch.manual_seed(0)
    model = moco.MoCo(opt).to(dev).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(5, 500, (8, 32), generator=g).to(dev)
    mask = torch.ones(8, 32, dtype=torch.bool, device=dev)
    for _ in range(2):                                   # moments and a key encoder that differs from the query encoder
        loss, _ = model(ids, mask, ids.flip(0), mask)
        loss.backward(); optimizer.step(); model.zero_grad()
    loss, _ = model(ids, mask, ids.flip(0), mask)
    loss.backward()
    eq, ek = model.encoder_q, model.encoder_k
    m_, v_ = optimizer._moments(eq)
    snap = [t.clone() for t in (eq.flat_params, ek.flat_params, m_, v_, eq.flat_grads())]
    steps0 = optimizer._steps

    def restore():
        with torch.no_grad():
            for dst, src in zip((eq.flat_params, ek.flat_params, m_, v_, eq.flat_grads()), snap):
                dst.copy_(src)
        eq.invalidate_mirror(); ek.invalidate_mirror()
        optimizer._steps = steps0
        model._ema_prefolded = None

    def run(fold, split):
        restore()
        optimizer._pre_tail = [lambda: None] if split else []
        model.allow_ema_fold(fold)
        optimizer.step()
        folded = model._ema_prefolded is not None
        model._momentum_update_key_encoder()
        torch.cuda.synchronize()
        return [t.clone() for t in (eq.flat_params, ek.flat_params, m_, v_, eq.mirror(), ek.mirror())], folded
    ref, f0 = run(False, False)
    assert not f0
    for split in (False, True):
        got, f1 = run(True, split)
        assert f1
        for a, b, what in zip(ref, got, ("query weights", "key weights", "m", "v", "query mirror", "key mirror")):
            assert torch.equal(a, b), (split, what)
    assert not torch.equal(ref[1], snap[1])              # the key encoder did move
    # a checkpoint between the folded step and the next forward is refused ...
    restore(); model.allow_ema_fold(True); optimizer._pre_tail = []
    optimizer.step()
    with pytest.raises(RuntimeError, match="folded"):
        utils.save(model, optimizer, scheduler, 3, opt, str(tmp_path), "bad")
    # ... and a torch-side write to an encoder (load_state_dict) cancels the skip: the next forward applies its EMA itself
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    before = ek.flat_params.clone()
    model._momentum_update_key_encoder()
    assert model._ema_prefolded is None and not torch.equal(ek.flat_params, before)
    model.allow_ema_fold(False)
