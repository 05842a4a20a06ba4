"""Answer matching for retrieved passages (the DPR recipe the reference uses, src/evaluation.py:25-117).

Host-side scoring of a finished retrieval: a passage "has the answer" when the lower-cased token sequence of any
answer string occurs contiguously in the passage's token sequence, both NFD-normalised; tokens are runs of letters /
digits / marks, or any other single character that is neither a separator nor a control.  `top_k_hits[i]` counts the
questions answered within the first i+1 passages.  Only what passage_retrieval.py calls is built (the reader metrics
and the retriever-inversion statistics of the reference file, :120-209, belong to other scripts)."""
import collections
import logging
import unicodedata
from functools import partial
from multiprocessing import Pool as ProcessPool
from typing import List

import regex

logger = logging.getLogger(__name__)

QAMatchStats = collections.namedtuple("QAMatchStats", ["top_k_hits", "questions_doc_hits"])


class SimpleTokenizer(object):
    ALPHA_NUM = r"[\p{L}\p{N}\p{M}]+"
    NON_WS = r"[^\p{Z}\p{C}]"

    def __init__(self):
        self._regexp = regex.compile("(%s)|(%s)" % (self.ALPHA_NUM, self.NON_WS),
                                     flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)

    def tokenize(self, text, uncased=False):
        tokens = [m.group() for m in self._regexp.finditer(text)]
        return [t.lower() for t in tokens] if uncased else tokens


def _normalize(text):
    return unicodedata.normalize("NFD", text)


def has_answer(answers, text, tokenizer) -> bool:
    words = tokenizer.tokenize(_normalize(text), uncased=True)
    for answer in answers:
        want = tokenizer.tokenize(_normalize(answer), uncased=True)
        n = len(want)
        if any(words[i:i + n] == want for i in range(len(words) - n + 1)):
            return True
    return False


def check_answer(example, tokenizer) -> List[bool]:
    hits = []
    for doc in example["ctxs"]:
        if doc["text"] is None:
            logger.warning("no doc in db")
            hits.append(False)
        else:
            hits.append(has_answer(example["answers"], doc["text"], tokenizer))
    return hits


def calculate_matches(data: List, workers_num: int):
    logger.info("Matching answers in top docs...")
    fn = partial(check_answer, tokenizer=SimpleTokenizer())
    if workers_num and workers_num > 1 and len(data) > 1:
        with ProcessPool(processes=min(workers_num, len(data))) as pool:
            scores = pool.map(fn, data)
    else:
        scores = [fn(ex) for ex in data]
    logger.info("Per question validation results len=%d", len(scores))
    n_docs = len(data[0]["ctxs"])
    top_k_hits = [0] * n_docs
    for question_hits in scores:
        best_hit = next((i for i, x in enumerate(question_hits) if x), None)
        if best_hit is not None:
            top_k_hits[best_hit:] = [v + 1 for v in top_k_hits[best_hit:]]
    return QAMatchStats(top_k_hits, scores)
