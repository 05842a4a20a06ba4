"""src/data.py against vectors captured from the reference's own src/data.py
(oracle/make_golden_data.py -> tests/golden/g11_data.json): same seeds, same samples, same batches."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

from src import data as D


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(os.path.dirname(__file__), "golden", "g11_data.json")) as f:
        return json.load(f)


def seed(n):
    random.seed(n)
    np.random.seed(n)


def tl(x):
    return [int(v) for v in (x.tolist() if torch.is_tensor(x) else x)]


def test_crop_specials_and_padding(g):
    c = g["randomcrop"]
    seed(c["seed"])
    x = torch.tensor(c["x"])
    assert [tl(D.randomcrop(x, *c["ratio"])) for _ in range(len(c["crops"]))] == c["crops"]
    x = torch.tensor(g["add_bos_eos"]["x"])
    for bos, eos, want in g["add_bos_eos"]["cases"]:
        assert tl(D.add_bos_eos(x, bos, eos)) == want
    assert tl(D.add_token(x, 9)) == g["add_token"]
    m = g["build_mask"]
    ids, mask = D.build_mask([torch.arange(3, 3 + n) for n in m["lens"]])
    assert ids.tolist() == m["ids"] and mask.int().tolist() == m["mask"]
    assert str(ids.dtype) == m["ids_dtype"] and str(mask.dtype) == m["mask_dtype"]


def test_augmentations_draw_in_reference_order(g):
    for name, fn in (("deleteword", lambda c: D.deleteword(list(c["x"]), p=c["p"])),
                     ("replaceword", lambda c: D.replaceword(list(c["x"]), c["lo"], c["hi"], p=c["p"])),
                     ("maskword", lambda c: D.maskword(list(c["x"]), c["mask_id"], p=c["p"])),
                     ("shuffleword", lambda c: D.shuffleword(list(c["x"]), p=c["p"]))):
        c = g[name]
        seed(c["seed"])
        assert tl(fn(c)) == c["y"], name


@pytest.mark.parametrize("aug", ["none", "delete", "shuffle"])
def test_dataset_and_collator_batches(g, aug):
    d = g["dataset"]
    tr = d["traces"][aug]
    tok = types.SimpleNamespace(bos_token_id=d["bos"], eos_token_id=d["eos"])
    opt = types.SimpleNamespace(ratio_min=0.1, ratio_max=0.5, augmentation=aug, prob_augmentation=0.2, chunk_length=d["chunk_length"],
                                mask_id=103, start_id=1000, vocab_size=2000)
    ds = D.Dataset([torch.tensor(x, dtype=torch.int) for x in d["docs"]], d["chunk_length"], tok, opt)
    seed(tr["seed"])
    items = [ds[i] for i in tr["order"]]
    assert [tl(it[0]["q_tokens"]) for it in items] == tr["q"]
    assert [tl(it[0]["k_tokens"]) for it in items] == tr["k"]
    batch, indices = D.Collator(opt)(items)
    assert list(indices) == tr["indices"]
    for key in ("q_tokens", "k_tokens"):
        assert batch[key].dtype == torch.int64 and batch[key].tolist() == tr[key]
    for key in ("q_mask", "k_mask"):
        assert batch[key].dtype == torch.bool and batch[key].int().tolist() == tr[key]


def test_multidataset_sampling(g):
    m = g["multidataset"]
    d = g["dataset"]
    tok = types.SimpleNamespace(bos_token_id=d["bos"], eos_token_id=d["eos"])
    opt = types.SimpleNamespace(ratio_min=0.1, ratio_max=0.5, augmentation="none", prob_augmentation=0.2, chunk_length=16)
    docs = [torch.tensor(x, dtype=torch.int) for x in d["docs"]]
    md = D.MultiDataset({"a": D.Dataset(docs[:4], 16, tok, opt), "b": D.Dataset(docs[4:], 16, tok, opt)})
    assert len(md) == m["len"]
    for c, want in m["probs"].items():
        md.set_prob(coeff=float(c))
        assert np.allclose(md.prob, want, rtol=1e-12)
    md.set_prob(coeff=m["coeff"])
    seed(m["seed"])
    for did, idx, q in m["picks"]:
        s, i = md[0]
        assert (s["dataset_id"], int(i), tl(s["q_tokens"])) == (did, idx, q)


def test_jsonl_loaders(tmp_path):
    class Tok:   # whitespace "tokenizer": ids = word lengths
        bos_token_id, eos_token_id = 1, 2

        def batch_encode_plus(self, lines, add_special_tokens=False):
            return {"input_ids": [[len(w) for w in l.split()] for l in lines]}

        def __call__(self, text, return_tensors="pt"):
            return {"input_ids": torch.tensor([[len(w) for w in text.split()]])}

    path = tmp_path / "c.jsonl"
    texts = ["a bb ccc dddd eeeee ffffff ggggggg hhhhhhhh", "x yy zzz", "one two three four five six seven eight nine ten"]
    path.write_text("".join(json.dumps({"text": t}) + "\n" for t in texts), encoding="utf-8")
    opt = types.SimpleNamespace(ratio_min=0.5, ratio_max=0.9, augmentation="none", prob_augmentation=0.0, chunk_length=6,
                                normalize_text=False, offsets_file=None, data_preprocessed=False, orig_sampling=False,
                                train_data=[str(path)], valid_data=[str(path)])
    docs = D.tokenize_jsonl_file(str(path), Tok(), opt)
    assert [tl(x) for x in docs] == [[len(w) for w in t.split()] for t in texts] and docs[0].dtype == torch.int
    lazy = D.LazyDataset(str(path), Tok(), opt)
    assert len(lazy) == 3
    seed(3)
    pair, idx = lazy[2]
    assert idx == 2 and int(pair["q_tokens"][0]) == 1 and int(pair["q_tokens"][-1]) == 2 and len(pair["q_tokens"]) <= 8
    train, val = D.load_data(opt, Tok(), is_main=True)
    assert len(train) == 3 and len(val) == 3
    batch, _ = D.Collator(opt)([train[i] for i in range(3)])
    assert batch["q_tokens"].shape[0] == 3 and batch["q_mask"].dtype == torch.bool
