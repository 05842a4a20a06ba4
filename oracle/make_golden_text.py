#!/usr/bin/env python3
"""Golden vectors for the two text helpers next to the retrieval scripts, captured from the reference's own
src/normalize_text.py (`normalize`) and src/evaluation.py (`has_answer`, `calculate_matches`).

TEST INFRASTRUCTURE ONLY; build container only (imports /root/reference).  Writes inputs and the reference's outputs as
JSON to tests/golden/g14_text.json; nothing of the reference's text is kept."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def import_reference():
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import src.evaluation as re_  # noqa
    import src.normalize_text as rn  # noqa
    assert re_.__file__.startswith(REF) and rn.__file__.startswith(REF)
    sys.path.remove(REF)
    return re_, rn


def main():
    re_, rn = import_reference()
    rng = random.Random(14)
    # every code point below U+3100 plus two further blocks, one per string with neighbours, and a few composed cases
    probes = ["a" + chr(c) + "b" for c in list(range(1, 0x3100)) + list(range(0xA780, 0xA790)) + list(range(0xFF00, 0xFF60))
              if not (0xD800 <= c <= 0xDFFF)]
    probes += ["wait . . . what", "x . . . y", "a…b … . . . ″‴", "it’s “quoted” ‘x’ — 1−2",
               "soft­hyphen", "tab\there\x0bvt\x0cff\x85nel", " .\x01 . . ", ""]
    alphabet = [chr(c) for c in (0x2d, 0x2010, 0x2212, 0x27, 0x2019, 0x60, 0xb4, 0x2032, 0x2033, 0x2026, 0x2f, 0x2044, 0x7e, 0x223c,
                                 0x20, 0x2e, 0x61, 0x159, 0x1, 0xad, 0x22, 0x201e)]
    probes += ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 40))) for _ in range(300)]
    norm = [[p, rn.normalize(p)] for p in probes]

    tok = re_.SimpleTokenizer()
    words = ["Karel", "IV.", "karla", "New", "York", "new-york", "žluťoučký", "kůň", "Příliš", "1348",
             "Praha", "praha,", "řeka", "Vltava", "U.S.A.", "e=mc²", "ﬁne", "İstanbul", "straße", "naïve",
             "café", "café", "東京", "(1999)", "—", "x_y"]
    cases = []
    for _ in range(400):
        text = " ".join(rng.choice(words) for _ in range(rng.randint(1, 14)))
        answers = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(1, 3))]
        if rng.random() < 0.3:
            answers.append(answers[0].upper())
        cases.append([answers, text, bool(re_.has_answer(answers, text, tok)), tok.tokenize(text, uncased=True)])
    data = []
    for qi in range(25):
        answers = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 2))) for _ in range(rng.randint(1, 2))]
        ctxs = [{"id": str(j), "title": "t",
                 "text": (None if rng.random() < 0.03 else " ".join(rng.choice(words) for _ in range(rng.randint(3, 12))))}
                for j in range(12)]
        data.append({"question": f"q{qi}", "answers": answers, "ctxs": ctxs})
    stats = re_.calculate_matches(data, 1)
    out = {"normalize": norm, "has_answer": cases,
           "matches": {"data": data, "top_k_hits": [int(v) for v in stats.top_k_hits],
                       "questions_doc_hits": [[bool(b) for b in row] for row in stats.questions_doc_hits]}}
    path = os.path.join(ROOT, "tests", "golden", "g14_text.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path) // 1000, "kB")


if __name__ == "__main__":
    main()
