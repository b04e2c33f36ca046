"""The initial topic assignments of addInstances (PTM:465-515) on CSR arrays, drawn with java.util.Random: the documented 48-bit
linear congruential generator (seed scramble, next(bits), nextInt(bound) with its power-of-two shortcut and its rejection loop), in
numpy.  Harness code for the synthetic workloads of bench.py, the tools and the full-size tests; a Java host draws them itself.

Draw order (PTM:466-515): entity by entity, view 0 .. M-1, position by position.  A token of view 0 takes nextInt(K) and its topic
joins the entity's list; a token of a later view takes list[nextInt(len(list))] -- with multiplicity -- or nextInt(K) when the entity
has no view-0 tokens.

The generator is sequential (s' = a*s + c mod 2^48) but linear, so a block of n consecutive states is s_j = A_j * s_0 + C_j with
A_j = a^j and C_j = c * (a^j - 1) / (a - 1), all mod 2^48 -- and the low 48 bits of a product are the low 48 bits of its wrap-around
in uint64.  A draw normally consumes one state; the rejection loop of nextInt consumes one more with probability (2^31 mod bound) /
2^31 (about 1e-7 at bound 400), which shifts every later draw by one state: those are found block by block and the block re-started
behind each.
"""
import numpy as np

_A = np.uint64(0x5DEECE66D)
_C = np.uint64(0xB)
_MASK = np.uint64((1 << 48) - 1)
_BLOCK = 1 << 20
_tables = {}


def _block_tables(n):
    """A_j, C_j for j = 1..n: state j steps after s is A_j * s + C_j (mod 2^48)."""
    if n not in _tables:
        a = np.empty(n, dtype=np.uint64)
        c = np.empty(n, dtype=np.uint64)
        # doubling: the tables of [1, 2m] from those of [1, m]: A_{m+j} = A_m * A_j, C_{m+j} = A_j * C_m + C_j
        a[0], c[0] = _A, _C
        m = 1
        with np.errstate(over="ignore"):
            while m < n:
                k = min(m, n - m)
                a[m:m + k] = (a[m - 1] * a[:k]) & _MASK
                c[m:m + k] = (a[:k] * c[m - 1] + c[:k]) & _MASK
                m += k
        _tables[n] = (a, c)
    return _tables[n]


def java_next_ints(seed, bounds):
    """java.util.Random(seed).nextInt(bounds[i]) for i = 0, 1, ... in order (bounds: positive int32 array)."""
    bounds = np.ascontiguousarray(bounds, dtype=np.uint32)             # (a bound is a positive int: below 2^31)
    out = np.empty(len(bounds), dtype=np.int32)
    s = np.uint64((int(seed) ^ 0x5DEECE66D) & ((1 << 48) - 1))
    A, Cc = _block_tables(_BLOCK)
    i, n = 0, len(bounds)
    with np.errstate(over="ignore"):
        while i < n:
            k = min(_BLOCK, n - i)
            st = (A[:k] * s + Cc[:k]) & _MASK                       # the k states after s
            u = (st >> np.uint64(17)).astype(np.uint32)              # next(31)
            b = bounds[i:i + k]
            r = u % b
            rej = (u - r) + (b - np.uint32(1)) >= np.uint32(1 << 31)   # int32 overflow of u - r + (bound - 1): draw again ...
            pow2 = (b & (b - np.uint32(1))) == 0
            if pow2.any():                                           # ... unless the bound is a power of two: the high bits, no loop
                r = np.where(pow2, ((b.astype(np.uint64) * u) >> np.uint64(31)).astype(np.uint32), r)
                rej &= ~pow2
            if rej.any():
                j = int(np.argmax(rej))                              # the first rejected draw: everything before it stands
                out[i:i + j] = r[:j]
                # draw i + j again from the following states, one at a time (rare), then go on from there
                s = st[j]
                bj = int(b[j])
                while True:
                    s = (_A * s + _C) & _MASK
                    uj = int(s >> np.uint64(17))
                    rj = uj % bj
                    if uj - rj + (bj - 1) < (1 << 31):
                        break
                out[i + j] = rj
                i += j + 1
            else:
                out[i:i + k] = r
                s = st[k - 1]
                i += k
    return out


def init_assignments(K, doc_off, seed):
    """z per view for the CSR offsets doc_off[m] ([D+1] each): the addInstances rule under java.util.Random(seed)."""
    M = len(doc_off)
    offs = [np.ascontiguousarray(o, dtype=np.int64) for o in doc_off]
    lens = [np.diff(o) for o in offs]
    per_doc = sum(lens)                                               # draws per entity
    total = int(per_doc.sum())
    run_start = np.concatenate([[0], np.cumsum(per_doc)[:-1]]).astype(np.int64)   # first draw of the entity's view-0 run
    l0 = lens[0]
    # draw index of every token of view m: the entity's run of that view starts at run_start, tokens follow in position order
    idx = []
    for m in range(M):
        n_m = int(offs[m][-1])
        ix = np.repeat(run_start - offs[m][:-1], lens[m])
        ix += np.arange(n_m, dtype=np.int64)
        idx.append(ix)
        run_start = run_start + lens[m]
    # bound of every draw: K for view 0 (and for a later view of an entity without view-0 tokens), else the length of its view-0 list
    bounds = np.full(total, K, dtype=np.uint32)
    b_ent = np.where(l0 > 0, l0, K).astype(np.uint32)
    for m in range(1, M):
        bounds[idx[m]] = np.repeat(b_ent, lens[m])
    draws = java_next_ints(seed, bounds)
    del bounds
    z0 = draws[idx[0]]
    z = [z0]
    for m in range(1, M):
        d = draws[idx[m]].astype(np.int64)
        has0 = np.repeat(l0 > 0, lens[m])
        base0 = np.repeat(offs[0][:-1], lens[m])
        pick = d.copy()
        if has0.any():
            pick[has0] = z0[(base0 + d)[has0]]
        z.append(pick.astype(np.int32))
    return z
