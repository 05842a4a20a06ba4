"""The asynchronous multi-GPU ordering, exercised on ONE GPU without any host synchronisation.

The gloo rehearsals (tests/test_gpu_dist.py) stage every collective through the host, which blocks the host on the side
stream at each of them and so serialises exactly the event orderings the data-parallel path relies on.  Here the
collectives are swapped for DEVICE-side stand-ins that run on whatever stream the product code put them on, behind a
long `torch.cuda._sleep` (so a consumer that does not wait for them is certain to read stale bytes), with no host sync
anywhere: the stand-in all-gather appends a fixed device-side function of this rank's rows as "rank 1", the stand-in
all-reduce averages the gradient slice with a fixed function of itself.  Every ordering of czech_contriever_amd.parallel
is then live: split_event before the first-half gather, keys_event before the second, head_event before the enqueue,
done_event before the next reader of the queue, the ranged all-reduces under the backward, wait_matrices before the
matrix AdamW and the tail all-reduce under it, the waits before an accumulating backward, the L2-from-init penalty and
clipping.  The same step sequence is then run again with every stand-in bracketed by device-wide synchronisation (the
serialised order).  Queue, pointer and weights must agree: bit for bit where the arithmetic is order-free, to the float
atomics' rounding elsewhere; a missed wait shows up as O(gradient) differences."""
import json
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moco_bert_oracle as O
from oracle.make_golden import TINY, ragged_batch

SLEEP = 4_000_000   # shader cycles in front of every stand-in collective (~2 ms): far longer than any kernel of the tiny model


def _opt(mdir, wdfi):
    return types.SimpleNamespace(
        queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.1, norm_doc=True, norm_query=True,
        moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=wdfi, retriever_model_id=mdir,
        pooling="average", random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98,
        eps=1e-6, warmup_steps=0, total_steps=10, lr_min_ratio=0.0, scheduler="linear")


def _install(monkeypatch, serial):
    """world = 2 everywhere; device-side stand-ins for the two collectives"""
    from czech_contriever_amd import parallel
    from src import dist_utils
    calls = {"gather": 0, "reduce": 0, "streams": set()}

    def fence():
        if serial:
            torch.cuda.synchronize()

    def all_gather_rows(out, x, group):
        fence()
        calls["gather"] += 1
        calls["streams"].add(torch.cuda.current_stream().cuda_stream)
        torch.cuda._sleep(SLEEP)
        n = x.shape[0]
        out[:n].copy_(x)
        out[n:].copy_(torch.nn.functional.normalize(x.flip(0) * 0.5 + 0.25, dim=1))    # "rank 1"
        fence()
        return out

    def average(flat, group=None):
        fence()
        calls["reduce"] += 1
        calls["streams"].add(torch.cuda.current_stream().cuda_stream)
        torch.cuda._sleep(SLEEP)
        flat.add_(flat.flip(0)).mul_(0.5)     # mean of g and a "rank 1" gradient: g reversed (NOT a multiple of g: AdamW's
                                              # update is invariant to the gradient's scale and would hide a stale read)
        fence()
        return flat

    monkeypatch.setattr(parallel, "world", lambda group=None: 2)
    monkeypatch.setattr(parallel, "_all_gather_rows", all_gather_rows)
    monkeypatch.setattr(parallel, "average_", average)
    monkeypatch.setattr(dist_utils, "get_world_size", lambda group=None: 2)
    return calls


def _run(mdir, monkeypatch, serial, split, update_freq, wdfi, clip, steps=3):
    from src import moco, utils
    from czech_contriever_amd.parallel import DataParallel
    calls = _install(monkeypatch, serial)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    opt = _opt(mdir, wdfi)
    P = O.make_params(O.BertShape(**TINY), 71)
    model = moco.MoCo(opt)
    model.key_gather_split = split
    model.encoder_q.load_state_dict(P)
    model.encoder_k.load_state_dict(P)
    if wdfi:
        model.init_weights = {k: v.clone() for k, v in model.encoder_q.state_dict().items()}
    with torch.no_grad():
        model.queue.copy_(torch.nn.functional.normalize(O.det_uniform((128, 64), 72), dim=0))
    model = model.to(dev).train()
    if wdfi:
        model.init_weights_to_gpu()
        with torch.no_grad():     # move the weights off their initial values, or the penalty and its gradient are exactly 0
            model.encoder_q.flat_params.add_(0.01 * torch.sin(torch.arange(model.encoder_q.flat_params.numel(), device=dev) * 0.37))
        model.encoder_q.invalidate_mirror()
    optimizer, scheduler = utils.set_optim(opt, model)
    wrapped = DataParallel(model, optimizer, chunk_layers=1)
    micro = 0
    for step in range(steps):
        for _ in range(update_freq):
            qt, qm = ragged_batch(8, 24, 9300 + micro, 512, [24, 17, 3, 24, 9, 24, 12, 20])
            kt, km = ragged_batch(8, 24, 9400 + micro, 512, [20, 24, 5, 11, 24, 7, 24, 16])
            micro += 1
            loss, _ = wrapped(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train")
            loss.backward()
        if clip is not None:
            optimizer.clip_grad_norm_(clip)
        optimizer.step()
        scheduler.step()
        wrapped.zero_grad()
    torch.cuda.synchronize()
    res = (model.encoder_q.flat_params.cpu(), model.encoder_k.flat_params.cpu(), model.queue.cpu(), int(model.queue_ptr))
    return res, calls


@pytest.mark.parametrize("split,update_freq,wdfi,clip", [(True, 1, False, None), (False, 1, False, None),
                                                         (True, 2, False, 0.05), (True, 2, True, None)])
def test_side_stream_orderings_without_host_sync(tmp_path, monkeypatch, split, update_freq, wdfi, clip):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mdir = str(tmp_path)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(dict(TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), f)
    kw = dict(split=split, update_freq=update_freq, wdfi=wdfi, clip=clip)
    (qa, ka, queue_a, ptr_a), calls = _run(mdir, monkeypatch, serial=False, **kw)
    (qs, ks, queue_s, ptr_s), _ = _run(mdir, monkeypatch, serial=True, **kw)
    steps, L = 3, TINY["num_hidden_layers"]
    main = torch.cuda.default_stream().cuda_stream
    # the collectives really ran off the main stream, the expected number of times: per micro-step one key exchange in
    # one or two pieces, and per backward one all-reduce per layer range (chunk_layers = 1) plus the vectors + embeddings tail
    assert main not in calls["streams"] and len(calls["streams"]) == 2
    assert calls["gather"] == steps * update_freq * (2 if split else 1)
    assert calls["reduce"] == steps * update_freq * (L + 1)
    assert ptr_a == ptr_s == (steps * update_freq * 16) % 64
    # step 0's keys come from identical weights in both runs: their queue columns are bit-identical, i.e. the gathered
    # rows (both pieces, rank-major order) reached the enqueue complete, and the enqueue ran after the head had read them
    assert torch.equal(queue_a[:, :16], queue_s[:, :16])
    assert float((queue_a - queue_s).abs().max()) < 1e-5
    for a, s in ((qa, qs), (ka, ks)):
        d = (a - s).abs()
        # float-atomic last bits through AdamW (a sign flip of a ~0 gradient moves a weight by ~lr): a race would move most weights
        assert float(d.max()) < 2.5e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))
