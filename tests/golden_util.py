"""Helpers shared by the golden / parity tests: materialise the table specs recorded in
tests/golden/goldens.json (the reference tests' INSERT statements restated as
generators) and a tiny text index builder for the BM25 cases."""
import json
import os
import re

import numpy as np

FLT_MAX = np.finfo(np.float32).max
_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goldens.json")


def load_goldens():
    with open(_GOLD) as f:
        return json.load(f)


def materialize(segs):
    """-> (ids int64[n], vectors f32[n, dim] (empty rows filled with FLT_MAX like
    MergeTreeVSManager.cpp:1380), empty bool[n])"""
    ids, vecs, empty = [], [], []
    for s in segs:
        dim = s["dim"]
        for n in range(s["start"], s["start"] + s["count"]):
            ids.append(n)
            kind = s["kind"]
            if kind == "empty":
                vecs.append([FLT_MAX] * dim)
                empty.append(True)
                continue
            empty.append(False)
            if kind == "nnn":
                vecs.append([float(n)] * dim)
            elif kind == "n_n3_n1":
                vecs.append([float(n), float(n + 3), float(n + 1)])
            elif kind == "cosine4":  # helpers/00000_prepare_index_cosine.sh, Float64 arithmetic then cast to Float32
                x = n / float(n) ** 2
                vecs.append([x, x, x, float(np.sqrt(1 - 3 * x ** 2))])
            elif kind == "mstg768":  # 00028_mqvs_index_mstg_build_search.sql
                vecs.append([0.00001 * (n * 768 + x + 1) * (-1 if x % 2 == 0 else 1) for x in range(768)])
            else:
                raise ValueError(kind)
    return np.array(ids, np.int64), np.array(vecs, np.float64).astype(np.float32), np.array(empty, bool)


def eval_filter(expr, ids):
    if expr is None:
        return np.ones(len(ids), bool)
    return np.array([bool(eval(expr, {"id": int(i)})) for i in ids], bool)


def f32_of(strings):
    return np.array([np.float32(s) for s in strings], np.float32)


# ------------------------------------------------------------------ text

def tokenize(text):
    """tantivy_search default tokenizer as pinned by 00040:
    lowercase runs of alphanumerics ("history's" -> history, s)."""
    return [t.lower() for t in re.findall(r"[A-Za-z0-9]+", text)]


class TextIndex:
    """Inverted index over one part: CSR postings (doc ids ascending), tf, fieldnorm ids.
    This is the flat-array export the BM25 C-ABI consumes (include/msvs.h, seam B)."""

    def __init__(self, docs_texts, fieldnorm_id):
        self.vocab = {}
        post = {}
        self.doc_len = []
        for doc, texts in enumerate(docs_texts):
            toks = []
            for t in texts:
                toks += tokenize(t)
            self.doc_len.append(len(toks))
            tf = {}
            for t in toks:
                tf[t] = tf.get(t, 0) + 1
            for t, c in tf.items():
                post.setdefault(t, []).append((doc, c))
        terms = sorted(post)
        self.vocab = {t: i for i, t in enumerate(terms)}
        self.post_off = np.zeros(len(terms) + 1, np.int64)
        doc_ids, tfs = [], []
        for i, t in enumerate(terms):
            for d, c in post[t]:
                doc_ids.append(d)
                tfs.append(c)
            self.post_off[i + 1] = len(doc_ids)
        self.doc_ids = np.array(doc_ids, np.uint32)
        self.tfs = np.array(tfs, np.uint32)
        self.fieldnorm_ids = np.array([fieldnorm_id(n) for n in self.doc_len], np.uint8)
        self.num_docs = len(docs_texts)
        self.total_tokens = int(sum(self.doc_len))

    def doc_freq(self, term):
        i = self.vocab.get(term)
        return 0 if i is None else int(self.post_off[i + 1] - self.post_off[i])
