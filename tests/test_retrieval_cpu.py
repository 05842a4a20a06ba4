"""CPU checks of the retrieval row's host logic: the index oracle against a brute-force loop, shard
arithmetic and index filling order, passage files, answer matching, rank discovery.  No GPU, no compute calls."""
import json
import os
import types

import numpy as np
import pytest

from czech_contriever_amd import retrieval
from oracle import flat_index_oracle


def test_index_oracle_is_the_definition():
    rng = np.random.default_rng(0)
    db = rng.integers(-3, 4, (60, 8)).astype(np.float32)
    q = rng.integers(-3, 4, (5, 8)).astype(np.float32)
    s, r = flat_index_oracle.search(db, q, 7)
    for i in range(5):
        scored = sorted(((-float(np.dot(q[i].astype(np.float64), db[j].astype(np.float64))), j) for j in range(60)))[:7]
        assert [j for _, j in scored] == r[i].tolist()              # descending score, ties by ascending row
        assert [-v for v, _ in scored] == s[i].astype(np.float64).tolist()
    s, r = flat_index_oracle.search(db[:3], q, 5)
    assert (r[:, 3:] == -1).all() and (s[:, 3:] == flat_index_oracle.FLT_LOWEST).all()
    sb, rb = flat_index_oracle.search_fp32_blas(db, q, 7)
    assert np.array_equal(sb, flat_index_oracle.search(db, q, 7)[0])


def test_shard_range_as_the_reference_splits():
    # generate_passage_embeddings.py:80-84: equal slices, the last shard takes the remainder
    assert [retrieval.shard_range(301, i, 2) for i in range(2)] == [(0, 150), (150, 301)]
    assert [retrieval.shard_range(10, i, 4) for i in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    assert retrieval.shard_range(7, 0, 1) == (0, 7)


def test_fill_index_slices_span_shards_in_order(tmp_path):
    sizes, paths, n0 = [250, 120, 30], [], 0
    for k, n in enumerate(sizes):
        ids = [f"id{n0 + i}" for i in range(n)]
        emb = np.arange(n0, n0 + n, dtype=np.float32)[:, None].repeat(4, 1)
        paths.append(str(tmp_path / f"passages_{k:02d}"))
        retrieval.write_shard(paths[-1], ids, emb)
        n0 += n

    class Recorder:
        def __init__(self):
            self.calls = []

        def index_data(self, ids, emb):
            self.calls.append((list(ids), np.array(emb)))

    rec = Recorder()
    retrieval.fill_index(rec, paths, 100)
    # the reference drains while MORE than indexing_batch_size rows are held (passage_retrieval.py:76-77), so the
    # last 100 rows wait for the final drain: 250 -> 100,100 | +120 = 170 -> 100 | +30 = 100 (held) -> 100
    assert [len(c[0]) for c in rec.calls] == [100, 100, 100, 100]
    assert sum((c[0] for c in rec.calls), []) == [f"id{i}" for i in range(400)]
    assert np.array_equal(np.concatenate([c[1] for c in rec.calls])[:, 0], np.arange(400, dtype=np.float32))
    rec2 = Recorder()
    retrieval.fill_index(rec2, paths[:1], 1000)
    assert [len(c[0]) for c in rec2.calls] == [250]


def test_passage_files_and_text(tmp_path):
    from src import data
    tsv = tmp_path / "p.tsv"
    tsv.write_text("id\ttext\ttitle\n7\tVltava je řeka\tVltava\n8\tBrno leží na Moravě\tBrno\n")
    got = data.load_passages(str(tsv))
    assert got == [{"id": "7", "title": "Vltava", "text": "Vltava je řeka"}, {"id": "8", "title": "Brno", "text": "Brno leží na Moravě"}]
    jl = tmp_path / "p.jsonl"
    jl.write_text(json.dumps({"id": 1, "text": "a"}) + "\n")
    assert data.load_passages(str(jl)) == [{"id": 1, "text": "a"}]
    assert data.load_passages(str(tmp_path / "missing.tsv")) is None
    assert retrieval.passage_text(got[0]) == "Vltava Vltava je řeka"
    assert retrieval.passage_text(got[0], no_title=True, lowercase=True) == "vltava je řeka"
    assert retrieval.passage_text({"text": "Bez Titulku"}) == "Bez Titulku"
    odd = {"title": "T\u2014x", "text": "it\u2019s \u201cq\u201d 1\u22122 \u2026"}
    assert retrieval.passage_text(odd, normalize=True) == "T-x it's \"q\" 1-2 ..."


def test_answer_matching():
    from src.evaluation import SimpleTokenizer, calculate_matches, has_answer
    t = SimpleTokenizer()
    assert t.tokenize("Karel IV., král (1346)") == ["Karel", "IV", ".", ",", "král", "(", "1346", ")"]
    assert has_answer(["karel iv."], "Za vlády KARLA IV. a pak Karel IV. zemřel", t)
    assert not has_answer(["new york"], "New-York city", t)          # tokens must be adjacent
    assert has_answer(["žluťoučký"], "Příliš žluťoučký kůň", t)   # NFD on both sides
    data = [{"answers": ["b c"], "ctxs": [{"text": "a b"}, {"text": "x B c y"}, {"text": None}]},
            {"answers": ["zzz"], "ctxs": [{"text": "a"}, {"text": "b"}, {"text": "c"}]},
            {"answers": ["a"], "ctxs": [{"text": "a"}, {"text": "b"}, {"text": "a"}]}]
    stats = calculate_matches(data, 1)
    assert stats.top_k_hits == [1, 2, 2]
    assert stats.questions_doc_hits == [[False, True, False], [False, False, False], [True, False, True]]


def test_contexts_and_recall_written_back():
    ex = [{"question": "q", "answers": ["řeka"]}]
    by_id = {"7": {"id": "7", "title": "Vltava", "text": "Vltava je řeka"}, "8": {"id": "8", "title": "Brno", "text": "město"}}
    retrieval.attach_contexts(ex, by_id, [(["8", "7"], np.array([2.5, 1.25], np.float32))])
    assert ex[0]["ctxs"] == [{"id": "8", "title": "Brno", "text": "město", "score": "2.5"},
                             {"id": "7", "title": "Vltava", "text": "Vltava je řeka", "score": "1.25"}]
    frac = retrieval.recall_report(ex, 1)
    assert frac == [0.0, 1.0] and [c["hasanswer"] for c in ex[0]["ctxs"]] == [False, True]
    with pytest.raises(ValueError):
        retrieval.attach_contexts(ex, by_id, [])


def test_rank_discovery(monkeypatch):
    from src import slurm
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLURM_JOB_ID", "SLURM_PROCID"):
        monkeypatch.delenv(k, raising=False)
    p = types.SimpleNamespace()
    slurm.init_distributed_mode(p)
    assert (p.global_rank, p.world_size, p.local_rank, p.is_distributed, p.is_main) == (0, 1, 0, False, True)
    monkeypatch.setenv("SLURM_JOB_ID", "1"); monkeypatch.setenv("SLURM_PROCID", "0")
    monkeypatch.setenv("SLURM_NTASKS", "1"); monkeypatch.setenv("SLURM_LOCALID", "0")
    p = types.SimpleNamespace()
    slurm.init_distributed_mode(p)
    assert p.world_size == 1 and p.is_main


def test_command_lines_keep_the_reference_flags():
    import generate_passage_embeddings as gen
    import passage_retrieval as ret
    a = gen.get_parser().parse_args([])
    assert (a.output_dir, a.prefix, a.shard_id, a.num_shards, a.per_gpu_batch_size, a.passage_maxlength) == \
        ("wikipedia_embeddings", "passages", 0, 1, 512, 512)
    assert not (a.no_fp16 or a.no_title or a.lowercase or a.normalize_text)
    b = ret.get_parser().parse_args(["--data", "x.jsonl"])
    assert (b.n_docs, b.validation_workers, b.per_gpu_batch_size, b.question_maxlength, b.indexing_batch_size,
            b.projection_size, b.n_subquantizers, b.n_bits, b.dataset) == (100, 32, 64, 512, 1000000, 768, 0, 8, "none")


def test_text_helpers_match_reference_golden(golden_dir):
    """src/normalize_text.normalize and src/evaluation.{SimpleTokenizer, has_answer, calculate_matches} against what the
    reference's own functions returned for the same inputs (g14, oracle/make_golden_text.py): every code point below
    U+3100, composed punctuation cases, 400 answer / passage pairs, one scored retrieval."""
    import json
    import os
    from src.evaluation import SimpleTokenizer, calculate_matches, has_answer
    from src.normalize_text import normalize
    with open(os.path.join(golden_dir, "g14_text.json")) as f:
        g = json.load(f)
    for text, want in g["normalize"]:
        assert normalize(text) == want, (text.encode("unicode_escape"), normalize(text), want)
    tok = SimpleTokenizer()
    for answers, text, want, tokens in g["has_answer"]:
        assert tok.tokenize(text, uncased=True) == tokens, text
        assert has_answer(answers, text, tok) == want, (answers, text)
    stats = calculate_matches(g["matches"]["data"], 1)
    assert list(stats.top_k_hits) == g["matches"]["top_k_hits"]
    assert [list(r) for r in stats.questions_doc_hits] == g["matches"]["questions_doc_hits"]
    assert calculate_matches(g["matches"]["data"], 2).top_k_hits == g["matches"]["top_k_hits"]
