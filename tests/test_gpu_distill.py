"""MI355X: the teacher/student distillation row — the similarity-matrix loss kernel, the MoCoDistill
student step and distill.py's command line (one process, and teacher + student as two ranks sharing the GPU over gloo)
against oracle/moco_bert_oracle.py."""
import json
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import moco_bert_oracle as O
from oracle.make_golden import TINY, ragged_batch


@pytest.mark.parametrize("B,D,Dt", [(32, 64, 96), (37, 50, 33), (256, 768, 3584)])
def test_sim_loss_kernel_matches_oracle(B, D, Dt):
    from czech_contriever_amd.ops import distill_sim_loss
    s = O.det_uniform((B, D), 11) * 3.0
    t = O.det_uniform((B, Dt), 12) + 0.1
    s_ref = s.clone().double().requires_grad_(True)
    want = O.distill_sim_loss(s_ref, t.double())
    want.backward()
    s_dev = s.cuda().requires_grad_(True)
    got = distill_sim_loss(s_dev, t.cuda())
    (got * 2.5).backward()
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-5 * max(1.0, abs(float(want.detach())))
    scale = float(s_ref.grad.abs().max())
    assert float((s_dev.grad.cpu().double() / 2.5 - s_ref.grad).abs().max()) <= 2e-5 * scale
    with torch.no_grad():                                  # loss only (no gradient wanted): same value
        assert abs(float(distill_sim_loss(s.cuda(), t.cuda())) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))


def _opt(mdir, **kw):
    base = dict(queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.0, norm_doc=True, norm_query=True,
                moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
                pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98,
                eps=1e-6, warmup_steps=0, total_steps=10, lr_min_ratio=0.0, scheduler="linear")
    base.update(kw)
    return types.SimpleNamespace(**base)


def _model_dir(tmp_path, vocab=False):
    d = str(tmp_path / "model")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), f)
    if vocab:
        with open(os.path.join(d, "vocab.txt"), "w") as f:
            f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(507)]) + "\n")
    return d


def test_student_step_matches_oracle(tmp_path):
    """MoCoDistill.forward returns (loss, q, stats) with q attached to the graph; the weighted sum of the contrastive
    and the similarity loss must give the oracle's gradients."""
    import distill
    from src import moco, utils
    dev = torch.device("cuda:0")
    mdir = _model_dir(tmp_path)
    opt = _opt(mdir)
    model = moco.MoCoDistill(opt)
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 71)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    queue = torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0)
    with torch.no_grad():
        model.queue.copy_(queue)
    model = model.to(dev).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    st = O.MoCoState(cfg, P, queue, momentum=0.99, temperature=0.05, norm_query=True, norm_doc=True)
    w = 0.3
    for step in range(2):
        qt, qm = ragged_batch(8, 24, 8300 + step, 512, [24, 17, 3, 24, 9, 24, 12, 20])
        kt, km = ragged_batch(8, 24, 8400 + step, 512, [20, 24, 5, 11, 24, 7, 24, 16])
        teacher = O.det_uniform((8, 40), 8500 + step)
        loss, q, stats = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), None, stats_prefix="train")
        assert q.requires_grad and q.shape == (8, TINY["hidden_size"])
        aux = distill.distill_loss(q, teacher.to(dev))
        ((1 - w) * loss + w * aux).backward()
        ref_loss, q_ref, _, ref_stats = st.forward_backward(qt, qm, kt, km, distill=(teacher, w, distill.DISTILL_LOSS_MULTIPLIER))
        assert abs(float(loss) - float(ref_loss)) < 3e-2 * max(1.0, abs(float(ref_loss)))
        assert abs(float(aux) - float(ref_stats["distill_loss"])) < 3e-2 * max(1.0, float(ref_stats["distill_loss"]))
        cosq = torch.nn.functional.cosine_similarity(q.detach().cpu(), q_ref, dim=1)
        assert float(cosq.min()) >= 1 - 1e-3
        gq = torch.cat([p.grad.reshape(-1) for p in model.encoder_q.parameters()]).cpu().double()
        gr = torch.cat([st.Pq[n].grad.reshape(-1) for n, _ in O.param_names(cfg)]).double()
        assert float(torch.nn.functional.cosine_similarity(gq, gr, dim=0)) > 0.995
        optimizer.step(); scheduler.step(); model.zero_grad()
        st.optimizer_step(1e-3)
    assert int(model.queue_ptr) == st.ptr == 16


CLI = ["--random_init", "--chunk_length", "32", "--per_gpu_batch_size", "8", "--queue_size", "64", "--projection_size", "128",
       "--temperature", "0.05", "--momentum", "0.99", "--lr", "1e-3", "--warmup_steps", "1", "--total_steps", "3",
       "--save_freq", "3", "--log_freq", "1", "--norm_query", "--norm_doc", "--distill_weight", "0.5"]


def test_distill_cli_one_process(tmp_path, monkeypatch):
    import distill
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    mdir = _model_dir(tmp_path)
    out = str(tmp_path / "run")
    distill.main(["--retriever_model_id", mdir, "--output_dir", out, "--save_dir", out, "--target_batch_size", "16",
                  "--teacher", "synthetic:48"] + CLI)
    ck = torch.load(os.path.join(out, "checkpoint", "step-3", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert ck["step"] == 3 and int(ck["model"]["queue_ptr"]) == (3 * 2 * 8) % 64   # update_freq 2
    assert torch.isfinite(ck["model"]["encoder_q.encoder.layer.0.output.dense.weight"]).all()
    # auto-resume from `latest` (step 3 = 6 micro-steps at update_freq 2): the dropout streams continue there instead of
    # replaying the masks of steps 0..2 (train.py does the same)
    seen = {}
    loop = distill.train_student

    def spy(opt, model, *a, **kw):
        m = model.module if hasattr(model, "module") else model
        seen.update(q=m.encoder_q.dropout_state(), k=m.encoder_k.dropout_state())
        return loop(opt, model, *a, **kw)
    monkeypatch.setattr(distill, "train_student", spy)
    distill.main(["--retriever_model_id", mdir, "--output_dir", out, "--save_dir", out, "--target_batch_size", "16",
                  "--teacher", "synthetic:48"] + [a if a != "3" else "5" for a in CLI])
    assert seen == {"q": {"salt": 0, "calls": 6}, "k": {"salt": 1, "calls": 6}}, seen
    monkeypatch.setattr(distill, "train_student", loop)
    # a BERT teacher from a checkpoint of this repo (decode -> re-tokenise -> encode on libczc)
    tdir = _model_dir(tmp_path / "t", vocab=True)
    torch.save({"opt": types.SimpleNamespace(retriever_model_id=tdir, pooling="average"), "step": 0,
                "model": {f"encoder_q.{k}": v for k, v in O.make_params(O.BertShape(**TINY), 5).items()}},
               os.path.join(tdir, "checkpoint.pth"))
    out2 = str(tmp_path / "run2")
    distill.main(["--retriever_model_id", tdir, "--output_dir", out2, "--save_dir", out2, "--target_batch_size", "8",
                  "--teacher", tdir] + CLI)
    assert os.path.exists(os.path.join(out2, "checkpoint", "step-3", "checkpoint.pth"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, mdir, out, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", CZC_DIST_BACKEND="gloo")
    try:
        import distill
        distill.main(["--retriever_model_id", mdir, "--output_dir", out, "--save_dir", out, "--target_batch_size", "8",
                      "--teacher", "synthetic:48"] + CLI)
        q.put((rank, "ok"))
    except BaseException as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))


def test_distill_cli_teacher_and_student_ranks(tmp_path):
    """rank 0 = teacher, rank 1 = student, both on this one GPU, collectives over gloo: the result must equal the
    one-process run with the same seed (the teacher's embeddings travel instead of being computed in place)."""
    import distill
    mdir = _model_dir(tmp_path)
    out = str(tmp_path / "two")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, mdir, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    solo = str(tmp_path / "solo")
    distill.main(["--retriever_model_id", mdir, "--output_dir", solo, "--save_dir", solo, "--target_batch_size", "8",
                  "--teacher", "synthetic:48"] + CLI)
    a = torch.load(os.path.join(out, "checkpoint", "step-3", "checkpoint.pth"), map_location="cpu", weights_only=False)
    b = torch.load(os.path.join(solo, "checkpoint", "step-3", "checkpoint.pth"), map_location="cpu", weights_only=False)
    assert a["step"] == b["step"] == 3
    assert torch.equal(a["model"]["queue_ptr"], b["model"]["queue_ptr"])
    wa, wb = a["model"]["encoder_q.encoder.layer.1.output.dense.weight"], b["model"]["encoder_q.encoder.layer.1.output.dense.weight"]
    # three Adam steps of 1e-3 on the same data: equal up to summation order inside a few kernels (LDS / global float atomics)
    assert float((wa - wb).abs().mean()) < 2e-6 and float((wa - wb).abs().max()) < 5e-4, (float((wa - wb).abs().mean()), float((wa - wb).abs().max()))


@pytest.mark.parametrize("tag", ["b8", "b64"])
def test_sim_loss_kernel_matches_reference_golden(golden_dir, tag):
    """czc_distill_sim_loss x DISTILL_LOSS_MULTIPLIER against what the reference's own compute_sim_matrix + mse x 100
    produced (tests/golden/g13_distill.npz, oracle/make_golden.py g_distill; distill.py:30,44-47,272-277)."""
    import numpy as np
    from czech_contriever_amd.ops import distill_sim_loss
    from tests.test_oracle_golden import distill_case
    g = dict(np.load(os.path.join(golden_dir, "g13_distill.npz")))
    student, teacher = distill_case(g, tag)
    s_dev = student.cuda().requires_grad_(True)
    aux = distill_sim_loss(s_dev, teacher.cuda()) * float(g["multiplier"])
    aux.backward()
    want = float(g[f"{tag}_loss"])
    assert abs(float(aux.detach()) - want) <= 2e-5 * abs(want), (float(aux.detach()), want)
    ref = torch.from_numpy(g[f"{tag}_dstudent"])
    assert float((s_dev.grad.cpu() - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
