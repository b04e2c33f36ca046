"""Deterministic synthetic multi-view corpora: no files, no
global RNG state — every draw is splitmix64(seed, stream, index).

K_true = K/2 latent topics.  Entity d: T_d = 1+Poisson(3) topics chosen
uniformly, weights theta_d ~ Dirichlet(1) shared by all views.  View v, latent
topic t: word = (r*A[v][t] + B[v][t]) mod V_v with rank r log-uniform over
[0, V_v) (~Zipf s=1) and A odd and coprime to V_v.  Side views are present
with probability 0.8 (an absent view is an empty CSR span = Assignments[m]==null,
MTA:19).  Lengths are 1+Poisson(lambda_v), or Pareto(1.5) truncated at 2048 for
the power-law config C5.
"""
from dataclasses import dataclass
from math import gcd

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def rand_u64(seed, stream, start, n):
    """n values of the counter-based generator for counters start..start+n-1."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))], dtype=np.uint64))[0]
        idx = np.arange(start, start + n, dtype=np.uint64)
        return _mix(base + (idx + np.uint64(1)) * _GOLD)


def rand_unit(seed, stream, start, n):
    return (rand_u64(seed, stream, start, n) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def rand_unit_at(seed, stream, idx):
    """Uniforms for arbitrary counters idx (int64 array)."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))], dtype=np.uint64))[0]
        x = _mix(base + (idx.astype(np.uint64) + np.uint64(1)) * _GOLD)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def _poisson_from_unit(u, lam):
    """Inverse-CDF Poisson (table up to lam + 12 sqrt(lam) + 20)."""
    kmax = int(lam + 12 * np.sqrt(lam) + 20)
    k = np.arange(kmax + 1, dtype=np.float64)
    from math import lgamma
    logp = -lam + k * np.log(lam) - np.array([lgamma(x + 1.0) for x in k])
    cdf = np.cumsum(np.exp(logp))
    cdf[-1] = 1.0
    return np.searchsorted(cdf, u, side="right").astype(np.int64)


@dataclass
class Corpus:
    K: int
    V: list                 # vocabulary size per view
    doc_off: list           # per view int64[D+1]
    tokens: list            # per view int32[N_m]
    name: str = ""

    @property
    def M(self):
        return len(self.V)

    @property
    def D(self):
        return len(self.doc_off[0]) - 1

    @property
    def total_tokens(self):
        return int(sum(int(o[-1]) for o in self.doc_off))

    def slice_docs(self, lo, hi):
        """Contiguous entity range [lo,hi) as its own corpus (document shards)."""
        offs, toks = [], []
        for m in range(self.M):
            o = self.doc_off[m]
            offs.append((o[lo:hi + 1] - o[lo]).astype(np.int64))
            toks.append(self.tokens[m][o[lo]:o[hi]].copy())
        return Corpus(self.K, list(self.V), offs, toks, f"{self.name}[{lo}:{hi}]")


def _doc_lengths(V, lo, hi, lam, seed, presence, power_law_text):
    """Per-view lengths of entities [lo,hi): depends only on (seed, entity id)."""
    n = hi - lo
    out = []
    for v in range(len(V)):
        pres = rand_unit(seed, 10 + v, lo, n) < presence[v]
        if power_law_text and v == 0:
            u = 1.0 - rand_unit(seed, 20 + v, lo, n)
            L = np.minimum(2048, np.floor(32.0 * u ** (-1.0 / 1.5))).astype(np.int64)
        else:
            L = 1 + _poisson_from_unit(rand_unit(seed, 20 + v, lo, n), float(lam[v]))
        out.append(np.where(pres, L, 0))
    return out


def doc_token_counts(K, V, D, lam, seed, presence=None, power_law_text=False):
    """Total tokens of every entity (all views) without generating the tokens: used to
    cut document shards balanced by token count."""
    M = len(V)
    if presence is None:
        presence = [1.0] + [0.8] * (M - 1)
    tot = np.zeros(D, dtype=np.int64)
    for c0 in range(0, D, 1 << 18):
        c1 = min(D, c0 + (1 << 18))
        for L in _doc_lengths(V, c0, c1, lam, seed, presence, power_law_text):
            tot[c0:c1] += L
    return tot


_POS_BITS = 12   # per-token draws are indexed by (entity << 12 | position); lengths are <= 2048


def generate(K, V, D, lam, seed, presence=None, power_law_text=False, chunk_docs=8192, name="",
             doc_lo=0, doc_hi=None):
    """Generate entities [doc_lo, doc_hi) of the D-entity corpus (default: all).  Every draw
    is a function of (seed, entity id, position), so any shard can be generated alone and
    equals the same slice of the full corpus.  lam[v] = Poisson mean of the length of view v."""
    M = len(V)
    if doc_hi is None:
        doc_hi = D
    if presence is None:
        presence = [1.0] + [0.8] * (M - 1)
    K_true = max(1, K // 2)
    # per (view, latent topic) affine permutations
    A = np.zeros((M, K_true), dtype=np.int64)
    B = np.zeros((M, K_true), dtype=np.int64)
    for v in range(M):
        a = (rand_u64(seed, 100 + v, 0, K_true) % np.uint64(V[v])).astype(np.int64)
        b = (rand_u64(seed, 200 + v, 0, K_true) % np.uint64(V[v])).astype(np.int64)
        for t in range(K_true):
            x = int(a[t]) | 1
            while gcd(x, V[v]) != 1:
                x += 2
            A[v, t] = x % V[v] if V[v] > 1 else 0
        B[v] = b
    nd = doc_hi - doc_lo
    doc_off = [np.zeros(nd + 1, dtype=np.int64) for _ in range(M)]
    tok_chunks = [[] for _ in range(M)]
    TMAX = 24
    for c0 in range(doc_lo, doc_hi, chunk_docs):
        c1 = min(doc_hi, c0 + chunk_docs)
        n = c1 - c0
        T = 1 + _poisson_from_unit(rand_unit(seed, 1, c0, n), 3.0)
        T = np.minimum(T, TMAX)
        tmax = int(T.max())
        # latent topics and Dirichlet(1) weights of each entity
        tt = (rand_u64(seed, 2, c0 * TMAX, n * TMAX).reshape(n, TMAX)[:, :tmax] % np.uint64(K_true)).astype(np.int64)
        e = -np.log(1.0 - rand_unit(seed, 3, c0 * TMAX, n * TMAX).reshape(n, TMAX)[:, :tmax])
        e[np.arange(tmax)[None, :] >= T[:, None]] = 0.0
        cum = np.cumsum(e, axis=1)
        cum /= cum[:, -1:]
        lens = _doc_lengths(V, c0, c1, lam, seed, presence, power_law_text)
        for v in range(M):
            L = lens[v]
            doc_off[v][c0 - doc_lo + 1:c1 - doc_lo + 1] = L
            nt = int(L.sum())
            if nt == 0:
                continue
            doc_of = np.repeat(np.arange(n), L)
            starts = np.cumsum(L) - L
            pos = np.arange(nt, dtype=np.int64) - np.repeat(starts, L)
            idx = ((doc_of.astype(np.int64) + c0) << _POS_BITS) | pos
            ut = rand_unit_at(seed, 30 + v, idx)
            j = (cum[doc_of] < ut[:, None]).sum(axis=1)
            j = np.minimum(j, T[doc_of] - 1)
            t = tt[doc_of, j]
            ur = rand_unit_at(seed, 40 + v, idx)
            r = np.floor(np.exp(ur * np.log(V[v] + 1.0))).astype(np.int64) - 1
            r = np.clip(r, 0, V[v] - 1)
            w = (r * A[v, t] + B[v, t]) % V[v]
            tok_chunks[v].append(w.astype(np.int32))
    tokens = []
    for v in range(M):
        np.cumsum(doc_off[v], out=doc_off[v])
        tokens.append(np.concatenate(tok_chunks[v]) if tok_chunks[v] else np.zeros(0, dtype=np.int32))
    return Corpus(K, list(V), doc_off, tokens, name)


# BASELINE.json configs; `scale` shrinks the entity count for tests.
CONFIGS = {
    "C2": dict(K=100, V=[20000], D=50_000, lam=[127], seed=0x5EED0002),
    "C3": dict(K=200, V=[50000, 5000, 5000], D=200_000, lam=[127, 7, 15], seed=0x5EED0003),
    "C4": dict(K=400, V=[50000, 5000, 5000], D=1_000_000, lam=[127, 7, 15], seed=0x5EED0004),
    # truncated HDP: the top 10 % of the topic ids start in inActiveTopicIndex (PTM:95, as optimizeDP
    # leaves topics no document holds, PTM:2450), alpha[m][K] > 0; initial assignments use the active topics only
    "C5": dict(K=1000, V=[50000, 5000, 5000, 5000, 5000], D=1_000_000, lam=[96, 7, 7, 7, 7],
               seed=0x5EED0005, power_law_text=True, inactive_from=900),
}


def config_inactive(name):
    """uint8[K] membership of inActiveTopicIndex at the start (None when the config has none) and the number of
    topics the initial assignments may use."""
    c = CONFIGS[name]
    if "inactive_from" not in c:
        return None, c["K"]
    ina = np.zeros(c["K"], dtype=np.uint8)
    ina[c["inactive_from"]:] = 1
    return ina, c["inactive_from"]


def make_config(name, D=None, doc_lo=0, doc_hi=None):
    c = dict(CONFIGS[name])
    if D is not None:
        c["D"] = int(D)
    return generate(c["K"], c["V"], c["D"], c["lam"], c["seed"],
                    power_law_text=c.get("power_law_text", False), name=name, doc_lo=doc_lo, doc_hi=doc_hi)


def config_doc_token_counts(name, D=None):
    c = dict(CONFIGS[name])
    if D is not None:
        c["D"] = int(D)
    return doc_token_counts(c["K"], c["V"], c["D"], c["lam"], c["seed"], power_law_text=c.get("power_law_text", False))


def shard_bounds(doc_tokens, n_shards):
    """Contiguous entity ranges balanced by token count."""
    cs = np.concatenate([[0], np.cumsum(doc_tokens)])
    total = cs[-1]
    cuts = [0]
    for r in range(1, n_shards):
        cuts.append(int(np.searchsorted(cs, total * r / n_shards, side="left")))
    cuts.append(len(doc_tokens))
    return [(cuts[r], cuts[r + 1]) for r in range(n_shards)]
