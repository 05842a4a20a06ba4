"""`src.data` for the MI355X path: the sampling / augmentation / collation pipeline that feeds
`train.py`, with the reference's names (Ajchler/czech-contriever src/data.py:27-57,105-207,423-652) so its callers and
configs carry over: `load_data`, `tokenize_jsonl_file`, `load_dataset_custom`, `LazyDataset`, `Dataset`,
`MultiDataset`, `Collator`, `randomcrop`, `build_mask`, `add_token`, `deleteword`, `replaceword`, `maskword`,
`shuffleword`, `apply_augmentation`, `add_bos_eos`.

This is host-side Python by nature (the reference's is too); the contract with the GPU path is the batch:
`q_tokens` / `k_tokens` int64 [B, Lq] / [B, Lk], `q_mask` / `k_mask` bool, pad id 0, each side padded to its own
batch maximum.  Every random decision draws from `random` / `numpy.random` in the reference's order (crop ratio, crop
start; one uniform vector per augmentation, one `randint` per replaced word, ...), so a seeded run reproduces the
reference's samples exactly: tests/test_data_cpu.py replays tests/golden/g11_data.json, captured from the reference.

Not built (outside §8): the distillation dataset, the `tokens.bin` corpus readers, passage loading."""
from __future__ import annotations

import glob
import json
import logging
import os
import pickle
import random
from collections import defaultdict

import numpy as np
import torch

from src import dist_utils

logger = logging.getLogger(__name__)


# ---- tokens in, tokens out ---------------------------------------------------------------------------------------
def randomcrop(x, ratio_min, ratio_max):
    """A contiguous span covering a U[ratio_min, ratio_max] fraction of x, at a uniform offset (two draws)."""
    share = random.uniform(ratio_min, ratio_max)
    n = int(len(x) * share)
    first = random.randint(0, len(x) - n)
    return x[first:first + n].clone()


def add_token(x, token):
    return torch.cat((torch.tensor([token]), x))


def add_bos_eos(x, bos_token_id, eos_token_id):
    if not torch.is_tensor(x):
        x = torch.Tensor(x)
    pieces = []
    if bos_token_id is not None:
        pieces.append(torch.tensor([bos_token_id]))
    pieces.append(x.clone().detach() if (bos_token_id is not None or eos_token_id is not None) else x)
    if eos_token_id is not None:
        pieces.append(torch.tensor([eos_token_id]))
    return pieces[0] if len(pieces) == 1 else torch.cat(pieces)


def build_mask(tensors):
    """Right-pad with id 0 to the longest sequence: (ids int64 [n, L], mask bool [n, L])."""
    width = max(len(t) for t in tensors)
    ids = torch.zeros(len(tensors), width, dtype=torch.long)
    mask = torch.zeros(len(tensors), width, dtype=torch.bool)
    for row, t in enumerate(tensors):
        ids[row, :len(t)] = t
        mask[row, :len(t)] = True
    return ids, mask


def deleteword(x, p=0.1):
    u = np.random.rand(len(x))
    return [tok for tok, keep in zip(x, u > p) if keep]


def replaceword(x, min_random, max_random, p=0.1):
    u = np.random.rand(len(x))
    return [tok if ui > p else random.randint(min_random, max_random) for tok, ui in zip(x, u)]


def maskword(x, mask_id, p=0.1):
    u = np.random.rand(len(x))
    return [tok if ui > p else mask_id for tok, ui in zip(x, u)]


def shuffleword(x, p=0.1):
    """Permute the values sitting at a random subset of positions (subset size ~ Binomial(len, p))."""
    k = int((np.random.rand(len(x)) < p).sum())
    where = random.sample(range(len(x)), k=k)
    moved = [x[i] for i in where]
    random.shuffle(moved)
    for i, v in zip(where, moved):
        x[i] = v
    return x


def apply_augmentation(x, opt):
    kind = opt.augmentation
    if kind == "mask":
        return torch.tensor(maskword(x, mask_id=opt.mask_id, p=opt.prob_augmentation))
    if kind == "replace":
        return torch.tensor(replaceword(x, min_random=opt.start_id, max_random=opt.vocab_size - 1, p=opt.prob_augmentation))
    if kind == "delete":
        return torch.tensor(deleteword(x, p=opt.prob_augmentation))
    if kind == "shuffle":
        return torch.as_tensor(shuffleword(x, p=opt.prob_augmentation))
    return x if torch.is_tensor(x) else torch.Tensor(x)


def _pair_from(tokens, opt, tokenizer):
    """query / key views of one chunk: crop, crop, then augmentation and specials per side, in that draw order."""
    q = randomcrop(tokens, opt.ratio_min, opt.ratio_max)
    k = randomcrop(tokens, opt.ratio_min, opt.ratio_max)
    q = add_bos_eos(apply_augmentation(q, opt), tokenizer.bos_token_id, tokenizer.eos_token_id)
    k = add_bos_eos(apply_augmentation(k, opt), tokenizer.bos_token_id, tokenizer.eos_token_id)
    return {"q_tokens": q, "k_tokens": k}


def _chunk(doc, chunk_length):
    first = random.randint(0, max(0, doc.size(0) - chunk_length))
    return doc[first:first + chunk_length]


# ---- datasets ------------------------------------------------------------------------------------------------------
class Dataset(torch.utils.data.Dataset):
    """Tokenised documents held in memory; an item is a (query, key) pair cut from one random chunk of one document."""

    def __init__(self, docs, chunk_length, tokenizer, opt):
        self.docs = docs
        self.chunk_length = chunk_length
        self.tokenizer = tokenizer
        self.opt = opt

    def __len__(self):
        return len(self.docs)

    def __getitem__(self, index):
        return _pair_from(_chunk(self.docs[index], self.chunk_length), self.opt, self.tokenizer), index

    def get_passage_from_all_docs(self):
        out = []
        for doc in self.docs:
            q = randomcrop(_chunk(doc, self.chunk_length), self.opt.ratio_min, self.opt.ratio_max)
            out.append(add_bos_eos(apply_augmentation(q, self.opt), self.tokenizer.bos_token_id, self.tokenizer.eos_token_id))
        return out


class LazyDataset(torch.utils.data.Dataset):
    """One jsonl file read line by line through a table of byte offsets (`opt.offsets_file`: a pickled list of ints, as
    the reference writes it; built on the fly when the option is empty)."""

    def __init__(self, path, tokenizer, opt):
        self.path = path
        self.tokenizer = tokenizer
        self.opt = opt
        self.chunk_length = opt.chunk_length
        offsets_file = getattr(opt, "offsets_file", None)
        if offsets_file:
            with open(offsets_file, "rb") as f:
                self.offsets = pickle.load(f)   # the user's own index file, written by their preprocessing
        else:
            self.offsets = self.generate_offset()

    def generate_offset(self):
        offsets, pos = [], 0
        with open(self.path, "rb") as f:
            for line in f:
                if line.strip():
                    offsets.append(pos)
                pos += len(line)
        return offsets

    def __len__(self):
        return len(self.offsets)

    def _create_pair(self, text):
        tokens = self.tokenizer(text, return_tensors="pt")["input_ids"].squeeze(0)
        return _pair_from(_chunk(tokens, self.chunk_length), self.opt, self.tokenizer)

    def __getitem__(self, index):
        with open(self.path, "r", encoding="utf-8") as f:
            f.seek(self.offsets[index])
            text = json.loads(f.readline())["text"]
        if getattr(self.opt, "normalize_text", False):
            text = _normalize(text)
        return self._create_pair(text), index


class MultiDataset(torch.utils.data.Dataset):
    """Several datasets behind one index space: every item first draws a dataset (numpy generator, probabilities from
    `set_prob`), then a uniform item of it (`random`); the index passed in is ignored, as in the reference."""

    def __init__(self, datasets):
        self.datasets = datasets
        self.dataset_ids = list(datasets.keys())
        self.prob = [1 / len(datasets) for _ in datasets]

    def __len__(self):
        return sum(len(d) for d in self.datasets.values())

    def __getitem__(self, index):
        which = np.random.choice(range(len(self.prob)), 1, p=self.prob)[0]
        did = self.dataset_ids[which]
        index = random.randint(0, len(self.datasets[did]) - 1)
        sample, _ = self.datasets[did][index]
        sample["dataset_id"] = did
        return sample, index

    def set_prob(self, coeff=0.0):
        share = np.array([float(len(d)) for d in self.datasets.values()])
        share /= share.sum()
        share = share ** coeff
        self.prob = share / share.sum()

    def all_docs(self):
        return [doc for d in self.datasets.values() for doc in d.docs]

    def get_passage_from_all_docs(self):
        return [p for d in self.datasets.values() for p in d.get_passage_from_all_docs()]


class Collator(object):
    def __init__(self, opt):
        self.opt = opt

    def __call__(self, batch_and_indices):
        examples, indices = zip(*batch_and_indices)
        batch = defaultdict(list)
        for ex in examples:
            for key, value in ex.items():
                batch[key].append(value)
        batch["q_tokens"], batch["q_mask"] = build_mask(batch["q_tokens"])
        batch["k_tokens"], batch["k_mask"] = build_mask(batch["k_tokens"])
        return batch, indices


# ---- loading -------------------------------------------------------------------------------------------------------
def _normalize(text):
    from src.normalize_text import normalize
    return normalize(text)


def tokenize_jsonl_file(file_path, tokenizer, opt):
    docs = []
    if file_path is None:
        return docs
    pending = []

    def flush():
        if pending:
            # (transformers 5 dropped batch_encode_plus: calling the tokenizer on a list is the same operation)
            encode = tokenizer.batch_encode_plus if hasattr(tokenizer, "batch_encode_plus") else tokenizer
            for ids in encode(pending, add_special_tokens=False)["input_ids"]:
                docs.append(torch.tensor(ids, dtype=torch.int))
            pending.clear()

    with open(file_path, "r", encoding="utf-8") as f:
        for line in f:
            if not line.strip():
                continue
            text = json.loads(line)["text"]
            pending.append(_normalize(text) if getattr(opt, "normalize_text", False) else text)
            if len(pending) > 100000:
                flush()
    flush()
    return docs


def load_dataset_custom(data_path, loading_mode):
    """Pre-tokenised shards `*.p*` (each a torch-saved list of int tensors) under data_path."""
    files = sorted(glob.glob(os.path.join(data_path, "*.p*")))
    if loading_mode == "split":
        files = list(np.array_split(files, dist_utils.get_world_size())[dist_utils.get_rank()])
    elif loading_mode == "single":
        files = files[:1]
    elif loading_mode != "full":
        files = []
    tensors = []
    for path in files:
        try:
            tensors.extend(torch.load(path, map_location="cpu", weights_only=True))
        except Exception:   # as the reference: a bad shard is skipped, not fatal
            logger.warning(f"Unable to load file {path}")
    return tensors or None


def load_data(opt, tokenizer, offsets=None, cumsums=None, is_main=False):
    """(train_dataset, val_dataset) as the reference's train.py expects them (src/data.py:105-137)."""
    def multi(paths):
        sets = {}
        for path in paths:
            docs = load_dataset_custom(path, opt.loading_mode)
            if docs is not None:
                sets[path] = Dataset(docs, opt.chunk_length, tokenizer, opt)
        out = MultiDataset(sets)
        out.set_prob(coeff=opt.sampling_coefficient)
        return out

    if getattr(opt, "data_preprocessed", False):
        return multi(opt.train_data), multi(opt.valid_data)
    if getattr(opt, "orig_sampling", False):
        raise NotImplementedError("--orig_sampling reads the tokens.bin corpus layout, which is outside the MI355X path")
    first = opt.train_data[0]
    if getattr(opt, "offsets_file", None) or os.path.getsize(first) > (1 << 30):
        train = LazyDataset(first, tokenizer, opt)
    else:
        train = Dataset(tokenize_jsonl_file(first, tokenizer, opt), opt.chunk_length, tokenizer, opt)
    val = None
    if is_main and getattr(opt, "valid_data", None):
        val = Dataset(tokenize_jsonl_file(opt.valid_data[0], tokenizer, opt), opt.chunk_length, tokenizer, opt)
    return train, val


def load_passages(path):
    """Passage table for encoding / retrieval (reference src/data.py:656-673): `.jsonl` lines as they are, otherwise a
    tab-separated file `id <TAB> text <TAB> title` whose header row (first field "id") is skipped."""
    import csv
    if not os.path.exists(path):
        logger.info(f"{path} does not exist")
        return
    logger.info(f"Loading passages from: {path}")
    passages = []
    with open(path) as fin:
        if path.endswith(".jsonl"):
            for line in fin:
                passages.append(json.loads(line))
        else:
            for row in csv.reader(fin, delimiter="\t"):
                if row[0] != "id":
                    passages.append({"id": row[0], "title": row[2], "text": row[1]})
    return passages
