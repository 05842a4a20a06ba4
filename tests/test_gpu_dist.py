"""Two ranks sharing ONE GPU over gloo (collectives staged through the host): the whole data-parallel MoCo
step (key all-gather in rank order, identical enqueue on both replicas, ranged backward + gradient averaging,
AdamW) must equal a single process that sees both ranks' batches concatenated.  RCCL itself needs one GPU
per rank; the collectives are the same torch.distributed calls."""
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _opt(mdir):
    return types.SimpleNamespace(
        queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.1, norm_doc=True, norm_query=True,
        moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=0, total_steps=10, lr_min_ratio=0.0, scheduler="linear")


def _batches(step):
    qt, qm = ragged_batch(8, 24, 9300 + step, 512, [24, 17, 3, 24, 9, 24, 12, 20])
    kt, km = ragged_batch(8, 24, 9400 + step, 512, [20, 24, 5, 11, 24, 7, 24, 16])
    return qt, qm, kt, km


def _run(mdir, rank, world, steps=3):
    from src import moco, utils
    from czech_contriever_amd.parallel import DataParallel
    dev = torch.device("cuda:0")
    opt = _opt(mdir)
    model = moco.MoCo(opt)
    P = O.make_params(O.BertShape(**TINY), 71)
    model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
    with torch.no_grad():
        model.queue.copy_(torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0))
    model = model.to(dev).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    wrapped = DataParallel(model, optimizer, chunk_layers=1) if world > 1 else model
    for step in range(steps):
        qt, qm, kt, km = _batches(step)
        if world > 1:   # each rank takes its half of the batch
            sl = slice(rank * 4, (rank + 1) * 4)
            qt, qm, kt, km = qt[sl], qm[sl], kt[sl], km[sl]
        loss, _ = wrapped(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train")
        loss.backward()
        optimizer.step(); scheduler.step(); wrapped.zero_grad()
    torch.cuda.synchronize()
    return (model.encoder_q.flat_params.cpu(), model.encoder_k.flat_params.cpu(), model.queue.cpu(), int(model.queue_ptr))


def _worker(rank, world, port, mdir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run(mdir, rank, world)
        out.put((rank, "ok", [t if isinstance(t, int) else t.numpy() for t in res]))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process_with_the_joint_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mdir = str(tmp_path)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(dict(TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), f)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mdir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    (q0, k0, queue0, ptr0), (q1, k1, queue1, ptr1) = res[0][2], res[1][2]
    # replicas stay bit-identical: same averaged gradients, same gathered keys in the same order
    assert (q0 == q1).all() and (k0 == k1).all() and (queue0 == queue1).all() and ptr0 == ptr1 == 24
    ref_q, ref_k, ref_queue, ref_ptr = _run(mdir, 0, 1)
    assert ref_ptr == 24
    assert float(abs(torch.from_numpy(queue0) - ref_queue).max()) < 2e-2      # keys are bf16-computed embeddings
    dq = torch.from_numpy(q0) - ref_q
    # AdamW's first steps move weights by ~lr = 1e-3 per step whatever the gradient scale: compare directions
    assert float(dq.abs().max()) < 2.5e-3, float(dq.abs().max())
    assert float(dq.abs().mean()) < 2e-4, float(dq.abs().mean())
