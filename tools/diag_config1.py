"""Diagnostic (GPU box): config-1-sized embeddings, HIP bf16 vs fp32 oracle -- how large is the sequence-specific part of
the pooled embeddings next to their common component, and how large is the HIP error next to it?"""
import json, os, sys, tempfile, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import moco_bert_oracle as O
from oracle.make_golden import ragged_batch
from src.contriever import Contriever, bert_config

shape = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
             max_position_embeddings=512, type_vocab_size=2)
cfg_o = O.BertShape(**shape)
P = O.make_params(cfg_o, 91)
enc = Contriever(bert_config(**shape))
enc.load_state_dict(P)
enc = enc.cuda().eval()
torch.set_num_threads(16)
for (B, S, lens) in [(4, 64, [64, 41, 64, 17]), (8, 64, [64] * 8)]:
    ids, mask = ragged_batch(B, S, 9300, 30522, lens)
    with torch.no_grad():
        ref, hid = O.encoder_forward(P, ids, mask, cfg_o, "average", False, return_hidden=True)
        got = enc(input_ids=ids.cuda(), attention_mask=mask.cuda()).cpu()
    c = ref.mean(0)
    eps_ref, eps_got = ref - c, got - c
    print(f"B={B} S={S}: |emb| {ref.norm(dim=1).tolist()}")
    print("  |common| %.3f  |specific| %s" % (float(c.norm()), [round(float(x), 3) for x in eps_ref.norm(dim=1)]))
    print("  |err| %s" % [round(float(x), 4) for x in (got - ref).norm(dim=1)])
    print("  cos(emb) %s" % [round(float(x), 6) for x in torch.nn.functional.cosine_similarity(got, ref, dim=1)])
    print("  cos(specific part) %s" % [round(float(x), 4) for x in torch.nn.functional.cosine_similarity(eps_got, eps_ref, dim=1)])
    G_ref, G_got = ref @ ref.t(), got @ got.t()
    print("  gram diff max %.3f (gram scale %.1f)" % (float((G_ref - G_got).abs().max()), float(G_ref.abs().max())))
    # per layer hidden-state error needs the workspace; instead: error of a 1-, 2-, 4-, 12-layer encoder
for L in (1, 2, 4, 8):
    sh = dict(shape, num_hidden_layers=L)
    co = O.BertShape(**sh)
    PL = {k: v for k, v in P.items() if not k.startswith("encoder.layer.") or int(k.split(".")[2]) < L}
    e = Contriever(bert_config(**sh)); e.load_state_dict(PL); e = e.cuda().eval()
    ids, mask = ragged_batch(4, 64, 9300, 30522, [64, 41, 64, 17])
    with torch.no_grad():
        ref = O.encoder_forward(PL, ids, mask, co, "average", False)
        got = e(input_ids=ids.cuda(), attention_mask=mask.cuda()).cpu()
    print(f"L={L}: |err| {[round(float(x),4) for x in (got-ref).norm(dim=1)]} |emb| {round(float(ref.norm(dim=1).mean()),2)} |specific| {[round(float(x),3) for x in (ref-ref.mean(0)).norm(dim=1)]}")
