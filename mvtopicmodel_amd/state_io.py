"""Reader of the text state `printState` writes (PTM:3276-3320; gzip when the name ends in .gz, PTM:3269-3274).

The reference's own reader, `initializeFromState`, is commented out (PTM:534-573); this one follows it -- skip the header,
then walk the entities in `data` order, their views in order, their positions in order, take column `topic` and insist
that column `typeindex` is the token at that position -- and also returns what the header holds: `gamma[m]*alpha[m][k]`
per view (PTM:3283) and `beta[0]`.  A row does not say which view it belongs to (`#doc source pos typeindex type topic`),
so the corpus decides: view m of entity d owns the next `len_m(d)` rows.  Columns are taken from the right (the source
string of an instance may contain blanks).

Python on purpose: it is harness (a JVM-less round trip for tests and tools), not a port of reference host code.
"""
import gzip

import numpy as np


def _open(path):
    path = str(path)
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def read_state(path, doc_off, tokens=None):
    """doc_off: per view, int64 [D+1] (the CSR of mvhdp_set_corpus); tokens: per view int32 [N_m] or None (then the
    typeindex column is not checked).  Returns dict(z=[per view int32 [N_m]], gamma_alpha=[M][K] float64, beta0=float)."""
    M = len(doc_off)
    D = len(doc_off[0]) - 1
    z = [np.full(int(doc_off[m][-1]), -1, dtype=np.int32) for m in range(M)]
    alpha_vals, beta0 = [[] for _ in range(M)], None
    with _open(path) as f:
        first = f.readline().rstrip("\n")
        if first != "#doc source pos typeindex type topic":
            raise ValueError("not a printState file: %r" % first[:60])
        cur = -1
        for line in f:                                      # '#alpha : modality:0' / values ... 'modality:1' / ... / '#beta[0] : b'
            line = line.rstrip("\n")
            if line.startswith("#beta[0] : "):
                beta0 = float(line[len("#beta[0] : "):])
                break
            for tok in line.replace("#alpha :", " ").split():
                if tok.startswith("modality:"):
                    cur = int(tok[len("modality:"):])
                    if not 0 <= cur < M:
                        raise ValueError("state has more views than the corpus")
                else:
                    alpha_vals[cur].append(float(tok))
        if beta0 is None:
            raise ValueError("state header ends without #beta[0]")
        for d in range(D):
            for m in range(M):
                b, e = int(doc_off[m][d]), int(doc_off[m][d + 1])
                for pos in range(e - b):
                    line = f.readline()
                    if not line:
                        raise ValueError("state ends early: entity %d view %d position %d" % (d, m, pos))
                    left, word, topic = line.rstrip("\n").rsplit(" ", 2)
                    left, type_index = left.rsplit(" ", 1)
                    left, p = left.rsplit(" ", 1)
                    doc = left.split(" ", 1)[0]
                    if int(doc) != d or int(p) != pos or (tokens is not None and int(type_index) != int(tokens[m][b + pos])):
                        raise ValueError("instance list and state do not match: " + line.rstrip("\n"))      # PTM:557-559
                    z[m][b + pos] = int(topic)
        if f.readline().strip():
            raise ValueError("state has more rows than the corpus has tokens")
    K = len(alpha_vals[0])
    if any(len(a) != K for a in alpha_vals):
        raise ValueError("state header: views with different numbers of topics")
    return dict(z=z, gamma_alpha=np.asarray(alpha_vals, dtype=np.float64), beta0=beta0)
