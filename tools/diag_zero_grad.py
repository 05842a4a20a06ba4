"""Diagnostic (GPU box): which gradient tensors are non-zero when the cotangent of the embeddings is exactly zero?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import moco_bert_oracle as O
from oracle.make_golden import ragged_batch, TINY
from src.contriever import Contriever, bert_config
from czech_contriever_amd import ops

for name, shape, B, S, lens in [
    ("tiny", TINY, 8, 24, [24, 17, 3, 24, 9, 24, 12, 20]),
    ("base2", dict(vocab_size=30522, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072, max_position_embeddings=512), 4, 64, [64, 41, 64, 17]),
    ("base2-full", dict(vocab_size=30522, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072, max_position_embeddings=512), 4, 64, [64] * 4)]:
    cfg_o = O.BertShape(**shape)
    P = O.make_params(cfg_o, 91)
    enc = Contriever(bert_config(**shape)); enc.load_state_dict(P); enc = enc.cuda().eval()
    ids, mask = ragged_batch(B, S, 9300, shape["vocab_size"], lens)
    emb = enc(input_ids=ids.cuda(), attention_mask=mask.cuda())
    (emb * 0.0).sum().backward()
    torch.cuda.synchronize()
    print(f"== {name}: zero cotangent -> |flat grad| {float(enc.flat_grads().norm()):.4e}")
    for n, p in enc.named_parameters():
        v = float(p.grad.abs().max())
        if v != 0.0 or not bool(torch.isfinite(p.grad).all()):
            print(f"   {n}: max |g| {v:.4e}")
    # and the head at these magnitudes
    q = emb.detach().clone().requires_grad_(True)
    k = (emb.detach() * 1.0).clone()
    K = 512
    queue = O.init_queue(shape["hidden_size"], K, 0).cuda()
    qb = torch.empty(K, shape["hidden_size"], dtype=torch.bfloat16, device="cuda")
    ops.queue_import(queue, qb)
    ws = ops.infonce_workspace(B, shape["hidden_size"], K, torch.device("cuda:0"))
    lse, stats = ops.infonce_forward(q.detach(), k, qb, 0.05, 0.0, ws)
    dq = ops.infonce_backward(q.detach(), k, qb, 0.05, 0.0, lse, 1.0, ws)
    lpos = (q.detach() * k).sum(1) * (1.0 / 0.05)
    print(f"   head: loss {float(stats[0]):.6f} |dq| {float(dq.norm()):.4e} max {float(dq.abs().max()):.4e}; lse - z0 {[(float(a) - float(b)) for a, b in zip(lse, lpos)]}")
