#!/usr/bin/env python3
"""BASELINE config C1: SampleData/SMSSpamCollection2.txt, one view, K = 20 -- CPU plumbing.

The reference's driver for this config (PTMFlow) has no source in the reference and there is no JVM here, so the
corpus goes through this loader, which restates what MALLET's text pipes do to a line `id<TAB>label<TAB>text`
(cc.mallet.pipe.CharSequenceLowercase, then cc.mallet.pipe.SimpleTokenizer as read from the 2.0.8 class file: letters
(Character.getType 1-5), marks (6-8) and '_' extend a token; space / line / paragraph separators and the punctuation
types 20-24, 29, 30 end it; every other character -- digits, controls, symbols -- is skipped WITHOUT ending the token;
a token in the stoplist is dropped), plus the build's own rule of dropping tokens shorter than 3 characters.  No
reference test pins the tokenisation: vocabulary parity with a Java run is NOT claimed.  What C1 checks is the plumbing:
text -> alphabet -> CSR -> addInstances rule -> sweeps -> counts / log-likelihood / top words.

  python tools/c1_smsspam.py --make-fixture      # /root/reference -> tests/golden/c1_smsspam.npz (integers + vocabulary)
  python tools/c1_smsspam.py --iterations 50     # restated reference (oracle/ref_threaded.c) on the fixture: LL/token, top words

Measurement / test tooling: it uses the oracle, it is not product code.
"""
import argparse
import os
import sys
import unicodedata

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
FIXTURE = os.path.join(ROOT, "tests", "golden", "c1_smsspam.npz")

_EXTEND = {"Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Me", "Mc"}
_END = {"Zs", "Zl", "Zp", "Pd", "Ps", "Pe", "Pc", "Po", "Pi", "Pf"}


def simple_tokenize(text, stop, min_len=3):
    """CharSequenceLowercase + SimpleTokenizer.pipe, then the stoplist and the minimum length."""
    out, cur = [], []

    def flush():
        if cur:
            tok = "".join(cur)[:1000]
            if tok not in stop and len(tok) >= min_len:
                out.append(tok)
            cur.clear()
    for ch in text.lower():
        cat = unicodedata.category(ch)
        if cat in _EXTEND or ch == "_":
            cur.append(ch)
        elif cat in _END:
            flush()
        # everything else (digits, control characters, symbols) is skipped without breaking the token
    flush()
    return out


def load(path=os.path.join(REF, "SampleData", "SMSSpamCollection2.txt"), stoplist=os.path.join(REF, "stoplists", "en.txt")):
    stop = set()
    with open(stoplist, encoding="utf-8", errors="replace") as f:
        for line in f:
            stop.update(line.split())
    vocab, names, labels, docs = {}, [], [], []
    with open(path, encoding="utf-8", errors="replace") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t", 2)
            if len(parts) < 3:
                continue
            toks = simple_tokenize(parts[2], stop)
            names.append(parts[0]); labels.append(parts[1])
            docs.append([vocab.setdefault(t, len(vocab)) for t in toks])       # Alphabet: first-seen order
    doc_off = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    tokens = np.asarray([t for d in docs for t in d], dtype=np.int32)
    words = np.asarray(sorted(vocab, key=vocab.get))
    return doc_off, tokens, words, np.asarray(names), np.asarray(labels)


def load_fixture():
    z = np.load(FIXTURE, allow_pickle=False)
    return z["doc_off"], z["tokens"], z["vocab"]


def run_reference_port(doc_off, tokens, words, K=20, iterations=50, threads=4, seed=1, quiet=False):
    """The restated reference (threads, queues, live updates) on the corpus; returns LL/token per 10 iterations,
    the final counts and the top words per topic (PTM:1792-1811 ordering: count descending, ties by descending id)."""
    from oracle.binding import Oracle
    from mvtopicmodel_amd.native import Hyper
    V = [len(words)]
    hy = Hyper.defaults(K, V)
    o = Oracle(K, V)
    o.set_corpus(0, doc_off, tokens)
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
    o.init_assignments(seed)
    o.build_counts()
    n = float(doc_off[-1])
    lls = [(0, float(o.model_log_likelihood()[0] / n))]
    done = 0
    while done < iterations:
        step = min(10, iterations - done)
        o.threaded_estimate(threads, step, seed + done)
        done += step
        lls.append((done, float(o.model_log_likelihood()[0] / n)))
        if not quiet:
            print(f"iteration {done}: LL/token {lls[-1][1]:.4f}", flush=True)
    nwk, nk = o.get_counts(0)
    z = o.get_assignments(0)
    top = []
    for k in range(K):
        order = sorted(range(V[0]), key=lambda w: (-int(nwk[w, k]), -w))[:8]
        top.append([str(words[w]) for w in order if nwk[w, k] > 0])
    o.close()
    return lls, nwk, nk, z, top


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-fixture", action="store_true")
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--threads", type=int, default=4)
    args = ap.parse_args()
    if args.make_fixture:
        doc_off, tokens, words, names, labels = load()
        np.savez_compressed(FIXTURE, doc_off=doc_off, tokens=tokens, vocab=words.astype("U"))
        print(f"{FIXTURE}: {len(doc_off) - 1} documents, {len(tokens)} tokens, {len(words)} types")
        return
    doc_off, tokens, words = load_fixture()
    print(f"C1: {len(doc_off) - 1} documents, {len(tokens)} tokens, {len(words)} types, K=20 "
          f"(vocabulary parity with a Java run not claimed)")
    lls, nwk, nk, z, top = run_reference_port(doc_off, tokens, words, iterations=args.iterations, threads=args.threads)
    for k, ws in enumerate(top):
        print(f"topic {k:2d} ({int(nk[k]):6d} tokens): " + " ".join(ws))


if __name__ == "__main__":
    main()
