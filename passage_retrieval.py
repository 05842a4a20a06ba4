#!/usr/bin/env python3
"""passage_retrieval.py — question encoding + exact top-k passage search on MI355X behind the reference's command line.

    python passage_retrieval.py --model_name_or_path <dir> --passages psgs.tsv \\
        --passages_embeddings "wikipedia_embeddings/passages_*" --data nq-test.jsonl --output_dir out --n_docs 100

Flags and files as Ajchler/czech-contriever passage_retrieval.py:148-249: shards from generate_passage_embeddings.py
go into the index --indexing_batch_size rows at a time (or `index.faiss` is reused with --save_or_load_index), every
file matched by --data is rewritten under --output_dir with a `ctxs` list (id, title, text, score, hasanswer) per
question, and the R@k line is printed.  The index lives in HBM and `search_knn` is libczc.so's fp32 matrix-core scan
with an exact running top-k (src/index.py, czech_contriever_amd/csrc/flat_index.hip)."""
import argparse
import glob
import json
import os
import time

import src.contriever
import src.data
import src.index
import src.slurm
from czech_contriever_amd import retrieval

os.environ["TOKENIZERS_PARALLELISM"] = "true"

FLAGS = [
    ("--data", dict(required=True, type=str, default=None, help=".json/.jsonl with `question` and `answers` per entry (glob)")),
    ("--passages", dict(type=str, default=None, help="Path to passages (.tsv file)")),
    ("--passages_embeddings", dict(type=str, default=None, help="Glob path to encoded passages")),
    ("--output_dir", dict(type=str, default=None, help="Results are written to outputdir with data suffix")),
    ("--n_docs", dict(type=int, default=100, help="Number of documents to retrieve per questions")),
    ("--validation_workers", dict(type=int, default=32, help="Number of parallel processes to validate results")),
    ("--per_gpu_batch_size", dict(type=int, default=64, help="Batch size for question encoding")),
    ("--save_or_load_index", dict(action="store_true", help="If enabled, save index and load index if it exists")),
    ("--model_name_or_path", dict(type=str, help="directory holding checkpoint.pth")),
    ("--no_fp16", dict(action="store_true", help="fp32 question embeddings")),
    ("--question_maxlength", dict(type=int, default=512, help="Maximum number of tokens in a question")),
    ("--indexing_batch_size", dict(type=int, default=1000000, help="Batch size of the number of passages indexed")),
    ("--projection_size", dict(type=int, default=768)),
    ("--n_subquantizers", dict(type=int, default=0, help="0 = flat index (the only one built on this path)")),
    ("--n_bits", dict(type=int, default=8, help="Number of bits per subquantizer")),
    ("--lang", dict(nargs="+")),
    ("--dataset", dict(type=str, default="none")),
    ("--lowercase", dict(action="store_true", help="lowercase text before encoding")),
    ("--normalize_text", dict(action="store_true", help="clean up dashes, quotes, controls before tokenising (src/normalize_text.py)")),
]


def embed_queries(args, queries, model, tokenizer):
    texts = [retrieval.prepare_text(q, args.lowercase, args.normalize_text) for q in queries]
    emb = retrieval.encode_texts(model.eval(), tokenizer, texts, args.per_gpu_batch_size, args.question_maxlength)
    print(f"Questions embeddings shape: {emb.size()}")
    return emb.cpu().numpy()


def build_index(args):
    index = src.index.Indexer(args.projection_size, args.n_subquantizers, args.n_bits)
    shards = sorted(glob.glob(args.passages_embeddings))
    home = os.path.dirname(shards[0])
    if args.save_or_load_index and os.path.exists(os.path.join(home, "index.faiss")):
        index.deserialize_from(home)
        return index
    print(f"Indexing passages from files {shards}")
    t0 = time.time()
    retrieval.fill_index(index, shards, args.indexing_batch_size)
    print(f"Indexing time: {time.time()-t0:.1f} s.")
    if args.save_or_load_index:
        index.serialize(home)
    return index


def main(args):
    print(f"Loading model from: {args.model_name_or_path}")
    model, tokenizer, _ = src.contriever.load_retriever(args.model_name_or_path)
    model = model.eval().cuda()
    if not args.no_fp16:
        model = model.half()
    index = build_index(args)
    by_id = {p["id"]: p for p in src.data.load_passages(args.passages)}

    for path in glob.glob(args.data):
        examples = retrieval.read_questions(path)
        q = embed_queries(args, [ex["question"] for ex in examples], model, tokenizer)
        t0 = time.time()
        found = index.search_knn(q, args.n_docs)
        print(f"Search time: {time.time()-t0:.1f} s.")
        retrieval.attach_contexts(examples, by_id, found)
        retrieval.recall_report(examples, args.validation_workers)
        target = os.path.join(args.output_dir, os.path.basename(path))
        os.makedirs(os.path.dirname(target), exist_ok=True)
        with open(target, "w") as out:
            for ex in examples:
                out.write(json.dumps(ex, ensure_ascii=False) + "\n")
        print(f"Saved results to {target}")


def get_parser():
    parser = argparse.ArgumentParser()
    for flag, kw in FLAGS:
        parser.add_argument(flag, **kw)
    return parser


if __name__ == "__main__":
    cli = get_parser().parse_args()
    src.slurm.init_distributed_mode(cli)
    main(cli)
