#!/usr/bin/env python3
"""generate_passage_embeddings.py — passage encoding on MI355X behind the reference's command line.

    python generate_passage_embeddings.py --model_name_or_path <dir with checkpoint.pth> --passages psgs.tsv \\
        --output_dir wikipedia_embeddings --shard_id 0 --num_shards 1 --per_gpu_batch_size 512

Same flags, same shard arithmetic and the same output file (`<output_dir>/<prefix>_<shard:02d>`, pickle of
`(ids, embeddings [n, H])`) as Ajchler/czech-contriever generate_passage_embeddings.py:69-124.  The work itself is
czech_contriever_amd.retrieval.encode_texts over libczc.so's encoder forward: bf16 matrix cores with fp32
accumulation from fp32 master weights whatever the flags say; `--no_fp16` only decides whether the stored
embeddings are fp32 or (default, like the reference's `model.half()`) fp16."""
import argparse
import os

import src.contriever
import src.data
import src.slurm
from czech_contriever_amd import retrieval

FLAGS = [
    ("--passages", dict(type=str, default=None, help="Path to passages (.tsv file)")),
    ("--output_dir", dict(type=str, default="wikipedia_embeddings", help="dir path to save embeddings")),
    ("--prefix", dict(type=str, default="passages", help="prefix path to save embeddings")),
    ("--shard_id", dict(type=int, default=0, help="Id of the current shard")),
    ("--num_shards", dict(type=int, default=1, help="Total number of shards")),
    ("--per_gpu_batch_size", dict(type=int, default=512, help="Batch size for the passage encoder forward pass")),
    ("--passage_maxlength", dict(type=int, default=512, help="Maximum number of tokens in a passage")),
    ("--model_name_or_path", dict(type=str, help="directory holding checkpoint.pth")),
    ("--no_fp16", dict(action="store_true", help="store fp32 embeddings")),
    ("--no_title", dict(action="store_true", help="title not added to the passage body")),
    ("--lowercase", dict(action="store_true", help="lowercase text before encoding")),
    ("--normalize_text", dict(action="store_true", help="clean up dashes, quotes, controls before tokenising (src/normalize_text.py)")),
]


def embed_passages(args, passages, model, tokenizer):
    """-> (ids, float ndarray [n, H]) for a list of passage dicts (the reference's function of the same name, :26-67)."""
    texts = [retrieval.passage_text(p, args.no_title, args.lowercase, args.normalize_text) for p in passages]
    emb = retrieval.encode_texts(model, tokenizer, texts, args.per_gpu_batch_size, args.passage_maxlength, progress_every=100000)
    return [p["id"] for p in passages], emb.cpu().numpy()


def main(args):
    model, tokenizer, _ = src.contriever.load_retriever(args.model_name_or_path)
    print(f"Model loaded from {args.model_name_or_path}.", flush=True)
    model = model.eval().cuda()
    if not args.no_fp16:
        model = model.half()
    passages = src.data.load_passages(args.passages)
    lo, hi = retrieval.shard_range(len(passages), args.shard_id, args.num_shards)
    print(f"Embedding generation for {hi - lo} passages from idx {lo} to {hi}.")
    ids, emb = embed_passages(args, passages[lo:hi], model, tokenizer)
    os.makedirs(args.output_dir, exist_ok=True)
    target = os.path.join(args.output_dir, f"{args.prefix}_{args.shard_id:02d}")
    print(f"Saving {len(ids)} passage embeddings to {target}.")
    retrieval.write_shard(target, ids, emb)
    print(f"Total passages processed {len(ids)}. Written to {target}.")


def get_parser():
    parser = argparse.ArgumentParser()
    for flag, kw in FLAGS:
        parser.add_argument(flag, **kw)
    return parser


if __name__ == "__main__":
    cli = get_parser().parse_args()
    src.slurm.init_distributed_mode(cli)
    main(cli)
