import json, os, sys, tempfile, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import moco_bert_oracle as O
from oracle.make_golden import TINY, ragged_batch
from src import moco, utils
from czech_contriever_amd import engine
dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
json.dump(dict(TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), open(os.path.join(d, "config.json"), "w"))
opt = types.SimpleNamespace(queue_size=64, momentum=0.99, temperature=0.05, label_smoothing=0.0, norm_doc=False, norm_query=False,
    moco_train_mode_encoder_k=False, weight_decay=0.01, weight_decay_from_init=False, retriever_model_id=d, pooling="average",
    random_init=True, projection_size=128, optim="adamw", lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, warmup_steps=2, total_steps=10,
    lr_min_ratio=0.0, scheduler="linear")
model = moco.MoCo(opt).to(dev).train()
optimizer, scheduler = utils.set_optim(opt, model)
B, S = 8, 24
qt, qm = ragged_batch(B, S, 7300, 512, [S, 17, 3, 24, 9, 24, 12, 20])
kt, km = ragged_batch(B, S, 7400, 512, [20, S, 5, 11, 24, 7, 24, 16])
orig = engine._EncoderFn.backward
def dbg(ctx, d_emb):
    print("encoder backward called, |d_emb| =", float(d_emb.abs().sum()))
    return orig(ctx, d_emb)
engine._EncoderFn.backward = staticmethod(dbg)
loss, stats = model(qt.to(dev), qm.to(dev), kt.to(dev), km.to(dev), stats_prefix="train")
print("loss", float(loss.detach()), "requires_grad", loss.requires_grad, loss.grad_fn)
loss.backward()
g = model.encoder_q.flat_grads(create=False)
print("flat grad:", None if g is None else float(g.abs().sum()))
print("p.grad none count:", sum(p.grad is None for p in model.encoder_q.parameters()))
