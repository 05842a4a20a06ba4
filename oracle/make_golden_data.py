#!/usr/bin/env python3
"""Golden vectors for the data path, captured from the reference's own src/data.py.

TEST INFRASTRUCTURE ONLY.  Imports the reference package from /root/reference (build container only), drives its
sampling / augmentation / collation helpers under fixed `random` and `numpy.random` seeds, and writes the inputs and
outputs as plain JSON to tests/golden/g11_data.json.  The build's src/data.py consumes the two generators in the same
order, so the same seeds must reproduce these outputs exactly (tests/test_data_cpu.py)."""
import argparse
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def import_reference_data():
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import src.data as rd  # noqa
    assert rd.__file__.startswith(REF), rd.__file__
    sys.path.remove(REF)
    return rd


def seed(n):
    random.seed(n)
    np.random.seed(n)


def tl(x):
    return [int(v) for v in (x.tolist() if torch.is_tensor(x) else x)]


def main():
    rd = import_reference_data()
    out = {}
    base = torch.arange(1000, 1040)
    seed(1)
    out["randomcrop"] = {"x": tl(base), "ratio": [0.1, 0.5], "seed": 1,
                         "crops": [tl(rd.randomcrop(base, 0.1, 0.5)) for _ in range(6)]}
    x = torch.tensor([5, 6, 7])
    out["add_bos_eos"] = {"x": tl(x), "cases": [[b, e, tl(rd.add_bos_eos(x, b, e))] for b, e in ((101, 102), (None, 102), (101, None), (None, None))]}
    out["add_token"] = tl(rd.add_token(x, 9))
    rag = [torch.arange(3, 3 + n) for n in (5, 1, 8, 3)]
    ids, mask = rd.build_mask(rag)
    out["build_mask"] = {"lens": [5, 1, 8, 3], "ids": ids.tolist(), "mask": mask.int().tolist(), "ids_dtype": str(ids.dtype), "mask_dtype": str(mask.dtype)}
    words = list(range(200, 230))
    seed(2); out["deleteword"] = {"x": words, "p": 0.3, "seed": 2, "y": tl(rd.deleteword(list(words), p=0.3))}
    seed(3); out["replaceword"] = {"x": words, "p": 0.3, "lo": 1000, "hi": 2000, "seed": 3, "y": tl(rd.replaceword(list(words), 1000, 2000, p=0.3))}
    seed(4); out["maskword"] = {"x": words, "p": 0.3, "mask_id": 103, "seed": 4, "y": tl(rd.maskword(list(words), 103, p=0.3))}
    seed(5); out["shuffleword"] = {"x": words, "p": 0.3, "seed": 5, "y": tl(rd.shuffleword(list(words), p=0.3))}

    tok = types.SimpleNamespace(bos_token_id=101, eos_token_id=102)
    docs = [torch.arange(10 * i, 10 * i + n, dtype=torch.int) for i, n in enumerate((40, 7, 25, 64, 3), start=1)]
    traces = {}
    for aug in ("none", "delete", "shuffle"):
        opt = types.SimpleNamespace(ratio_min=0.1, ratio_max=0.5, augmentation=aug, prob_augmentation=0.2, chunk_length=16,
                                    mask_id=103, start_id=1000, vocab_size=2000)
        ds = rd.Dataset([d.clone() for d in docs], 16, tok, opt)
        seed(10)
        items = [ds[i] for i in (0, 3, 1, 2, 4, 3)]
        batch, indices = rd.Collator(opt)(items)
        traces[aug] = {"seed": 10, "order": [0, 3, 1, 2, 4, 3],
                       "q": [tl(it[0]["q_tokens"]) for it in items], "k": [tl(it[0]["k_tokens"]) for it in items],
                       "q_tokens": batch["q_tokens"].tolist(), "q_mask": batch["q_mask"].int().tolist(),
                       "k_tokens": batch["k_tokens"].tolist(), "k_mask": batch["k_mask"].int().tolist(), "indices": list(indices)}
    out["dataset"] = {"docs": [tl(d) for d in docs], "chunk_length": 16, "bos": 101, "eos": 102, "traces": traces}

    opt = types.SimpleNamespace(ratio_min=0.1, ratio_max=0.5, augmentation="none", prob_augmentation=0.2, chunk_length=16)
    md = rd.MultiDataset({"a": rd.Dataset([d.clone() for d in docs[:4]], 16, tok, opt), "b": rd.Dataset([d.clone() for d in docs[4:]], 16, tok, opt)})
    probs = {}
    for c in (0.0, 0.5, 1.0):
        md.set_prob(coeff=c)
        probs[str(c)] = [float(p) for p in md.prob]
    md.set_prob(coeff=0.5)
    seed(11)
    picks = []
    for _ in range(8):
        s, idx = md[0]
        picks.append([s["dataset_id"], int(idx), tl(s["q_tokens"])])
    out["multidataset"] = {"sizes": {"a": 4, "b": 1}, "probs": probs, "seed": 11, "coeff": 0.5, "picks": picks, "len": len(md)}

    path = os.path.join(ROOT, "tests", "golden", "g11_data.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
