"""Answer matching for retrieved passages: the DPR recipe the reference's retrieval script scores with
(reference src/evaluation.py:25-117, called from passage_retrieval.py:95-108).

A passage "has the answer" when the token sequence of any answer string occurs contiguously in the passage's token
sequence, both NFD-normalised and lower-cased; a token is a run of letters / digits / combining marks, or any single
character that is neither a separator nor a control.  `calculate_matches` returns, for every rank r, how many questions
were answered within their first r + 1 passages, plus the per-question hit lists.  Pinned against the reference's own
functions on shared inputs (tests/golden/g14_text.json).  Only what passage_retrieval.py calls is built here (the reader
metrics and retriever-inversion statistics of the reference file, :120-209, serve other scripts)."""
import collections
import logging
import unicodedata
from multiprocessing import Pool

import numpy as np
import regex

logger = logging.getLogger(__name__)

QAMatchStats = collections.namedtuple("QAMatchStats", ["top_k_hits", "questions_doc_hits"])

_TOKEN = regex.compile(r"[\p{L}\p{N}\p{M}]+|[^\p{Z}\p{C}]", flags=regex.IGNORECASE | regex.UNICODE | regex.MULTILINE)
_SEP = "\x00"   # a control character: never part of a token


class SimpleTokenizer:
    """`tokenize(text, uncased=False) -> list[str]` (the one method of the reference's tokenizer that the scoring uses)."""

    def tokenize(self, text, uncased=False):
        found = _TOKEN.findall(text)
        return [w.lower() for w in found] if uncased else found


def _sequence(tokenizer, text):
    """token sequence as one string, every token fenced by separators: sub-sequence search becomes substring search"""
    return _SEP + _SEP.join(tokenizer.tokenize(unicodedata.normalize("NFD", text), uncased=True)) + _SEP


def has_answer(answers, text, tokenizer=None) -> bool:
    tokenizer = tokenizer or SimpleTokenizer()
    haystack = _sequence(tokenizer, text)
    for answer in answers:
        needle = _sequence(tokenizer, answer)
        if needle == _SEP + _SEP:          # an answer without tokens matches anywhere (the empty slice equals itself)
            return True
        if needle in haystack:
            return True
    return False


def _question_hits(example):
    tokenizer = SimpleTokenizer()
    hits = []
    for passage in example["ctxs"]:
        if passage["text"] is None:
            logger.warning("no doc in db")
            hits.append(False)
        else:
            hits.append(has_answer(example["answers"], passage["text"], tokenizer))
    return hits


def calculate_matches(data, workers_num):
    logger.info("Matching answers in top docs...")
    if workers_num and workers_num > 1 and len(data) > 1:
        with Pool(processes=min(workers_num, len(data))) as pool:
            per_question = pool.map(_question_hits, data)
    else:
        per_question = [_question_hits(example) for example in data]
    logger.info("Per question validation results len=%d", len(per_question))
    depth = len(data[0]["ctxs"])
    first = np.zeros(depth + 1, dtype=np.int64)           # histogram of the rank of each question's first hit
    for hits in per_question:
        first[hits.index(True) if True in hits else depth] += 1
    return QAMatchStats(np.cumsum(first[:depth]).tolist(), per_question)
