"""The bench's EXACT workload inside pytest -m gpu (BASELINE.json configs[1]: bert-base-sized encoder, chunk_length 256,
per_gpu_batch_size 256, queue_size 131 072, temperature 0.05, momentum 0.9995, dropout 0.1 on in both encoders): the
M = 65 536-row products (9-12 output tiles walked per workgroup), 3 072 attention items, 36-way split-K weight gradients and
the 131 072-slot head otherwise run only inside bench.py, whose one check is a finite loss.  Here:
  * two identical micro-steps from one state are compared bit for bit (loss, weight-matrix gradients, enqueued queue columns;
    vector / embedding gradients within the float-atomic bound DESIGN.md §4 (9) states), everything finite, queue_ptr advanced
    by 256;
  * the forward of that batch -- both encoders in TRAINING mode, every dropout site on -- against the fp32 CPU oracle under
    the same dropout outcomes: pooled embeddings (cosine >= 1 - 1e-3, north_star's bar) and the InfoNCE loss;
  * seeded 40-case slices of tools/fuzz_gemm.py and tools/fuzz_attention.py (random shapes between the pinned ones).
Model, options and batches come from bench.py itself, so this cannot drift from what is measured."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import bench
from oracle import moco_bert_oracle as O  # the checker, never the product
from tests import dropout_ref as D

B, S, K = 256, 256, 131072


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def headline(gpu, tmp_path_factory):
    """bench.py's model at its default flags (seed 0) and its first synthetic batch of rank 0."""
    from src import moco
    mdir = str(tmp_path_factory.mktemp("czc_headline"))
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(bench.BERT_BASE, f)
    opt = bench.make_opt(mdir, B, 1, K)
    torch.manual_seed(opt.seed)
    from src import utils
    model = moco.MoCo(opt).to(gpu).train()
    optimizer, scheduler = utils.set_optim(opt, model)
    b0, batch = bench.synthetic_batches(0, B, S, 2, gpu)
    assert tuple(batch["q_tokens"].shape) == (B, S) and bool(batch["q_mask"].all())
    # bench.py's first step: with the initial queue (random unit columns) the positive dominates every row -- loss and gradient
    # are exactly 0, as in the reference.  It enqueues 256 keys, real negatives for the step under test,
    # and AdamW's first update makes the query encoder differ from the key encoder (the EMA is no longer a no-op).
    loss, _ = model(**b0, stats_prefix="train", iter_stats={})
    loss.backward()
    optimizer.step(); scheduler.step(); model.zero_grad()
    torch.cuda.synchronize()
    assert int(model.queue_ptr) == B
    return model, batch


def _snapshot(model):
    return (model.encoder_q.flat_params.clone(), model.encoder_k.flat_params.clone(), model.queue.clone(),
            model.queue_ptr.clone(), model.encoder_q.dropout_state(), model.encoder_k.dropout_state())


def _restore(model, snap):
    with torch.no_grad():
        model.encoder_q.flat_params.copy_(snap[0]); model.encoder_k.flat_params.copy_(snap[1])
        model.queue.copy_(snap[2]); model.queue_ptr.copy_(snap[3])
    model.encoder_q.invalidate_mirror(); model.encoder_k.invalidate_mirror()
    model.encoder_q.load_dropout_state(snap[4]); model.encoder_k.load_dropout_state(snap[5])
    model.zero_grad()


def test_headline_step_repeats_bitwise_and_is_finite(headline, gpu):
    model, batch = headline
    snap = _snapshot(model)
    n_mat = model.encoder_q._n_mat
    runs = []
    for _ in range(2):
        _restore(model, snap)
        loss, stats = model(**batch, stats_prefix="train", iter_stats={})
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), model.encoder_q.flat_grads().clone(), model.queue[:, B:2 * B].clone(),
                     int(model.queue_ptr), model.encoder_k.flat_params.clone(), float(stats["train/stdq"][0])))
    (l0, g0, c0, p0, k0, s0), (l1, g1, c1, p1, k1, s1) = runs
    assert torch.isfinite(l0) and torch.isfinite(g0).all() and torch.isfinite(c0).all()
    assert float(l0) > 0 and float(g0.abs().max()) > 0          # a real gradient, not the saturated case
    assert p0 == p1 == 2 * B
    assert torch.equal(l0, l1)
    assert abs(s0 - s1) <= 1e-6 * abs(s0)                         # the step statistics' column sums are float atomics
    assert torch.equal(k0, k1)                                    # the EMA
    assert torch.equal(c0, c1)                                    # the key encoder's forward and the enqueue
    assert not torch.equal(c0, snap[2][:, B:2 * B])               # ... which did overwrite slots 256 .. 511
    assert torch.equal(g0[:n_mat], g1[:n_mat])                    # split-K weight gradients: fixed summation order
    tail0, tail1 = g0[n_mat:], g1[n_mat:]                         # biases, LayerNorm, embeddings: fp32 atomics
    assert float((tail0 - tail1).abs().max()) <= 1e-5 * float(tail0.abs().max())
    # the weights AdamW is about to see are untouched by the forward / backward
    assert torch.equal(model.encoder_q.flat_params, snap[0])


def test_headline_batch_forward_vs_oracle_dropout_on(headline, gpu):
    """B 256 x S 256, both encoders in training mode: q / k embeddings and the loss against the oracle (no_grad on both
    sides; 2 x 11.75 TFLOP of fp32 on the host: about a minute on the box's 16 threads)."""
    model, batch = headline
    model.zero_grad()
    cfg = O.BertShape(**{k: v for k, v in bench.BERT_BASE.items() if k in O.BertShape.__dataclass_fields__})
    Pq = {n: v.detach().cpu().clone() for n, v in model.encoder_q.state_dict().items()}
    Pk = {n: v.detach().cpu().clone() for n, v in model.encoder_k.state_dict().items()}
    queue0 = model.queue.detach().cpu().clone()
    ptr0 = int(model.queue_ptr)
    sq, sk = model.encoder_q.dropout_seed(), model.encoder_k.dropout_seed()
    assert sq != sk
    with torch.no_grad():
        loss, q_dev, stats = model._step(batch["q_tokens"], batch["q_mask"], batch["k_tokens"], batch["k_mask"], None, "train", {})
    torch.cuda.synchronize()
    k_dev = model.queue[:, ptr0:ptr0 + B].t().cpu()               # the enqueue stores the fp32 keys as they are
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    ids_q, ids_k, m = batch["q_tokens"].cpu(), batch["k_tokens"].cpu(), batch["q_mask"].cpu()
    with torch.no_grad():
        q_ref = O.encoder_forward(Pq, ids_q, m, cfg, "average", False, training=True, masks=D.DeviceScales(sq, cfg, B, S, gpu))
        # the EMA precedes the key forward (src/moco.py:314); momentum 0.9995 on the oracle's copy of the key weights
        Pk = O.ema_update(Pk, Pq, 0.9995)
        k_ref = O.encoder_forward(Pk, ids_k, m, cfg, "average", False, training=True, masks=D.DeviceScales(sk, cfg, B, S, gpu))
        loss_ref, logits = O.infonce_loss(q_ref, k_ref, queue0, 0.05)
    cq = torch.nn.functional.cosine_similarity(q_dev.cpu().double(), q_ref.double(), dim=-1)
    ck = torch.nn.functional.cosine_similarity(k_dev.double(), k_ref.double(), dim=-1)
    assert float(cq.min()) >= 1 - 1e-3, (float(cq.min()), int(cq.argmin()))
    assert float(ck.min()) >= 1 - 1e-3, (float(ck.min()), int(ck.argmin()))
    # un-normalised embeddings at T = 0.05: the loss is a difference of logits of size ~1e3; bf16 operands leave ~1 % of it
    assert abs(float(loss) - float(loss_ref)) < 3e-2 * abs(float(loss_ref)) + 1.0, (float(loss), float(loss_ref))
    acc, stdq, stdk = O.step_stats(logits, q_ref, k_ref)
    assert abs(float(stats["train/accuracy"][0]) - float(acc)) <= 100.0 * 3 / B
    assert abs(float(stats["train/stdq"][0]) - float(stdq)) < 1e-2 * max(1.0, float(stdq))
    assert abs(float(stats["train/stdk"][0]) - float(stdk)) < 1e-2 * max(1.0, float(stdk))


def test_fuzz_gemm_slice(gpu):
    """40 seeded random GEMM problems of tools/fuzz_gemm.py: layouts, epilogues, ragged M / N / K, padded leading dimensions,
    split-K, every tile / K-loop setting, against an fp32 matmul."""
    from tools import fuzz_gemm
    done, fails = fuzz_gemm.run(40, seed=4, verbose=False)
    assert done == 40 and not fails, fails


def test_fuzz_attention_slice(gpu):
    """40 seeded random attention problems of tools/fuzz_attention.py: batch, length 1..512, heads, ragged padding, dropout
    on / off, keep words or regenerated draws; forward, dQ / dK / dV, bias sums, bitwise-repeatable backward."""
    from tools import fuzz_attention
    n, fails = fuzz_attention.run(40, seed=4, verbose=False)
    assert n == 40 and not fails, fails
