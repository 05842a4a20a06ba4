"""MI355X: the passage-encoding and retrieval command lines end to end on a tiny random encoder.

generate_passage_embeddings.py (two shards) -> passage_retrieval.py, checked against the oracles: the stored
embeddings against oracle/moco_bert_oracle.py's fp32 encoder on the same token ids (min cosine >= 1 - 1e-3, the
tolerance north_star states for pooled embeddings), and the retrieved ids / scores against
oracle/flat_index_oracle.py on the stored vectors."""
import json
import os
import pickle
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flat_index_oracle, moco_bert_oracle as O
from oracle.make_golden import TINY


def _setup(tmp_path):
    mdir = str(tmp_path / "model")
    os.makedirs(mdir, exist_ok=True)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(TINY, f)
    words = [f"w{i}" for i in range(507)]
    with open(os.path.join(mdir, "vocab.txt"), "w") as f:
        f.write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words) + "\n")
    cfg = O.BertShape(**TINY)
    P = O.make_params(cfg, 4242)
    opt = types.SimpleNamespace(retriever_model_id=mdir, pooling="average")
    torch.save({"opt": opt, "model": {f"encoder_q.{k}": v for k, v in P.items()}, "step": 0}, os.path.join(mdir, "checkpoint.pth"))
    rng = random.Random(7)
    passages = [{"id": str(100 + i), "title": " ".join(rng.choice(words) for _ in range(rng.randint(1, 3))),
                 "text": " ".join(rng.choice(words) for _ in range(rng.randint(5, 70)))} for i in range(301)]
    tsv = str(tmp_path / "psgs.tsv")
    with open(tsv, "w") as f:
        f.write("id\ttext\ttitle\n")
        for p in passages:
            f.write(f"{p['id']}\t{p['text']}\t{p['title']}\n")
    questions = [{"question": " ".join(rng.choice(words) for _ in range(rng.randint(3, 9))),
                  "answers": [passages[rng.randrange(301)]["text"].split()[0]]} for _ in range(11)]
    qfile = str(tmp_path / "questions.jsonl")
    with open(qfile, "w") as f:
        for q in questions:
            f.write(json.dumps(q) + "\n")
    return mdir, cfg, P, passages, tsv, questions, qfile


def test_encode_then_retrieve_command_lines(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import generate_passage_embeddings as gen
    import passage_retrieval as ret
    from src.contriever import load_retriever
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    mdir, cfg, P, passages, tsv, questions, qfile = _setup(tmp_path)
    emb_dir = str(tmp_path / "emb")
    for shard in (0, 1):
        a = gen.get_parser().parse_args(["--model_name_or_path", mdir, "--passages", tsv, "--output_dir", emb_dir,
                                         "--shard_id", str(shard), "--num_shards", "2", "--per_gpu_batch_size", "64",
                                         "--passage_maxlength", "48"])
        gen.main(a)
    ids, embs = [], []
    for shard in (0, 1):
        with open(os.path.join(emb_dir, f"passages_{shard:02d}"), "rb") as f:   # files this test just wrote
            i, e = pickle.load(f)
        ids += i
        embs.append(e)
    assert [len(e) for e in embs] == [150, 151] and ids == [p["id"] for p in passages]
    assert all(e.dtype == np.float16 for e in embs)              # model.half() in the reference -> fp16 shards
    emb = np.concatenate(embs).astype(np.float32)

    # embeddings against the fp32 oracle on the same token ids (title + " " + text, truncated to 48 tokens)
    _, tok, _ = load_retriever(mdir)
    texts = [p["title"] + " " + p["text"] for p in passages]
    enc = tok(texts, return_tensors="pt", max_length=48, padding=True, truncation=True)
    want = O.encoder_forward(P, enc["input_ids"], enc["attention_mask"].bool(), cfg, pooling="average").numpy()
    cos = (emb * want).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 1 - 1e-3, cos.min()

    out_dir = str(tmp_path / "out")
    a = ret.get_parser().parse_args(["--model_name_or_path", mdir, "--passages", tsv, "--passages_embeddings",
                                     os.path.join(emb_dir, "passages_*"), "--data", qfile, "--output_dir", out_dir,
                                     "--n_docs", "7", "--projection_size", str(TINY["hidden_size"]), "--per_gpu_batch_size", "4",
                                     "--indexing_batch_size", "100", "--validation_workers", "1", "--save_or_load_index"])
    ret.main(a)
    assert os.path.exists(os.path.join(emb_dir, "index.faiss")) and os.path.exists(os.path.join(emb_dir, "index_meta.faiss"))
    got = [json.loads(l) for l in open(os.path.join(out_dir, "questions.jsonl"))]
    assert [g["question"] for g in got] == [q["question"] for q in questions]

    model, tok, _ = load_retriever(mdir)
    model = model.eval().cuda().half()
    qemb = ret.embed_queries(a, [q["question"] for q in questions], model, tok).astype(np.float32)
    want_s, want_r = flat_index_oracle.search(emb, qemb, 7)
    by_id = {p["id"]: p for p in passages}
    for g, ws, wr in zip(got, want_s, want_r):
        assert [c["id"] for c in g["ctxs"]] == [ids[r] for r in wr]
        assert np.allclose([float(c["score"]) for c in g["ctxs"]], ws, rtol=0, atol=2e-5)
        for c in g["ctxs"]:
            assert c["text"] == by_id[c["id"]]["text"] and c["title"] == by_id[c["id"]]["title"]
            words_ = by_id[c["id"]]["text"].split()
            assert c["hasanswer"] == any(ans in words_ for ans in g["answers"])
    # second run reuses index.faiss
    ret.main(a)
    again = [json.loads(l) for l in open(os.path.join(out_dir, "questions.jsonl"))]
    assert again == got
