"""Diagnostic (GPU box): where does the config-1-sized step-1 loss leave the oracle's?"""
import json, os, sys, tempfile, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import moco_bert_oracle as O
from oracle.make_golden import ragged_batch
from src import moco, utils

shape = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
             max_position_embeddings=512, type_vocab_size=2)
dev = torch.device("cuda:0")
mdir = tempfile.mkdtemp()
json.dump(dict(shape, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), open(os.path.join(mdir, "config.json"), "w"))
opt = types.SimpleNamespace(
    queue_size=512, momentum=0.9995, temperature=0.05, label_smoothing=0.0, norm_doc=False, norm_query=False,
    moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=mdir,
    pooling="average", random_init=True, projection_size=768, optim="adamw", lr=1e-5, beta1=0.9, beta2=0.98,
    eps=1e-6, warmup_steps=0, total_steps=100, lr_min_ratio=1.0, scheduler="linear")
model = moco.MoCo(opt)
cfg_o = O.BertShape(**shape)
P = O.make_params(cfg_o, 91)
model.encoder_q.load_state_dict(P); model.encoder_k.load_state_dict(P)
queue = O.init_queue(768, 512, 0)
with torch.no_grad():
    model.queue.copy_(queue)
model = model.to(dev).train()
optimizer, scheduler = utils.set_optim(opt, model)
torch.set_num_threads(16)
st = O.MoCoState(cfg_o, P, queue, momentum=0.9995, temperature=0.05)
B, S = 4, 64
for step in range(2):
    qt, qm = ragged_batch(B, S, 9300 + step, 30522, [64, 41, 64, 17])
    kt, km = ragged_batch(B, S, 9400 + step, 30522, [52, 64, 9, 64])
    qpre = model.queue.detach().cpu().clone()
    loss, stats = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train", iter_stats={})
    loss.backward()
    g = model.encoder_q.flat_grads()
    print(f"step {step}: HIP loss {float(loss.detach()):.4f} acc {float(stats['train/accuracy'][0])} |grad| {float(g.norm()):.4e} max {float(g.abs().max()):.4e} finite {bool(torch.isfinite(g).all())}")
    w_before = model.encoder_q.flat_params.detach().clone()
    optimizer.step(); scheduler.step(); model.zero_grad()
    print(f"   |dW| max {float((model.encoder_q.flat_params - w_before).abs().max()):.3e}")
    queue_before = st.queue.clone()
    ref, q_ref, k_ref, rs = st.forward_backward(qt, qm, kt, km)
    gr = torch.cat([st.Pq[n].grad.reshape(-1) for n, _ in O.param_names(cfg_o)])
    print(f"   oracle loss {float(ref):.4f} acc {float(rs['accuracy'])} |grad| {float(gr.norm()):.4e}")
    st.optimizer_step(1e-5)
    # the same logits from torch on the device's own q, k and pre-step queue
    with torch.no_grad():
        qd = model.encoder_q(input_ids=qt.to(dev), attention_mask=qm.to(dev)).cpu()
    print("   q cos vs oracle", [round(float(x), 6) for x in torch.nn.functional.cosine_similarity(qd, q_ref, dim=1)], "(q after the optimizer step)")
    lo_ref, _ = O.infonce_loss(q_ref, k_ref, queue_before, 0.05)
    print(f"   oracle loss recomputed {float(lo_ref):.4f}; queue_pre diff HIP-oracle {float((qpre - queue_before).abs().max()):.4e}")
    qb = model._queue_mirror().float().cpu()       # [K, C] after enqueue
    print(f"   bf16 mirror vs fp32 queue after enqueue: {float((qb.t() - model.queue.cpu()).abs().max()):.4e}; queue vs oracle {float((model.queue.cpu() - st.queue).abs().max()):.4e}; ptr {int(model.queue_ptr)} {st.ptr}")
    kd_cols = model.queue.cpu()[:, step * B:(step + 1) * B].t()
    print("   enqueued keys cos vs oracle k", [round(float(x), 6) for x in torch.nn.functional.cosine_similarity(kd_cols, k_ref, dim=1)])
    lg = O.compute_logits(q_ref, k_ref, queue_before)
    print("   oracle logits: pos", [round(float(x), 2) for x in lg[:, 0]], "max neg", [round(float(x), 2) for x in lg[:, 1:].max(1).values], "argmax", lg[:, 1:].argmax(1).tolist())
