"""Known-answer tests pinning the CPU oracle (SURVEY.md §8c KAT-1..6 and the
documented third-party contracts).  CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding
from oracle.binding import JRand


def _tree(L, w):
    w = np.asarray(w, dtype=np.float64)
    t = np.zeros(2 * len(w), dtype=np.float64)
    L.orc_ftree_construct(t.ctypes.data, len(w), w.ctypes.data)
    return t


def test_kat1_ftree_power_of_two(oracle_lib):
    L = oracle_lib
    t = _tree(L, [1, 2, 3, 4])                      # FT:158-161
    assert list(t[1:]) == [10, 3, 7, 1, 2, 3, 4]
    s = lambda u: L.orc_ftree_sample(t.ctypes.data, 4, u)
    assert s(0.4) == 2 and s(0.0) == 0 and s(0.1) == 1 and s(0.29) == 1
    assert s(0.3) == 2        # 0.3*10 == 3.0 exactly; 3.0 < tree[2]=3 is false -> right child (Q8)
    assert s(1.0) == 3
    assert s(1.0000001) == -2  # IllegalArgumentException FT:112-114


def test_kat2_ftree_rotation(oracle_lib):
    L = oracle_lib
    t = _tree(L, [1, 2, 3, 4, 5])
    assert t[4] == 9 and t[3] == 5 and t[2] == 10 and t[1] == 15
    s = lambda x: L.orc_ftree_sample(t.ctypes.data, 5, x / 15.0)
    # in-order leaves are topics 3,4,0,1,2
    for x, k in [(0.5, 3), (3.9, 3), (4.5, 4), (8.9, 4), (9.5, 0), (10.5, 1), (11.9, 1), (12.5, 2), (14.9, 2)]:
        assert s(x) == k, (x, k)
    # general rule: rotation by P-K; K=400 -> leaves 112..399 then 0..111
    K = 400
    w = np.ones(K)
    t = _tree(L, w)
    order = [L.orc_ftree_sample(t.ctypes.data, K, (i + 0.5) / K) for i in range(K)]
    assert order == list(range(112, 400)) + list(range(0, 112))


def test_kat3_ftree_update(oracle_lib):
    L = oracle_lib
    w = np.array([0.5, 1.25, 2.0, 0.75, 3.5, 1.0, 0.25])
    t = _tree(L, w)
    root0 = t[1]
    L.orc_ftree_update(t.ctypes.data, len(w), 4, 1.5)
    assert t[len(w) + 4] == 1.5
    assert t[1] == root0 + (1.5 - 3.5)
    assert t[0] == (1.5 - 3.5) * 0 + t[0]   # tree[0] untouched by the loop's i>0 guard... (i/2 reaches 0 and stops)
    assert t[0] == 0.0


def test_kat4_lower_bound(oracle_lib):
    L = oracle_lib
    a = np.array([1.0, 3.0, 6.0])
    lb = lambda x: L.orc_lower_bound(a.ctypes.data, x, 3)
    assert [lb(0.5), lb(1.0), lb(1.0001), lb(6.0), lb(6.1)] == [0, 0, 1, 2, -1]


def test_kat5_thread_split():
    # numThreads=9 -> nst=6, nut=2 (PTM:1036-1037); delta for word w from worker t -> queue 6*(w%2)+t (WRK:589)
    T = 9
    nst, nut = 3 * T // 4, T // 4
    assert (nst, nut) == (6, 2)
    assert nst * (7 % nut) + 3 == 9
    assert [1 * nst + st for st in range(nst)] == list(range(6, 12))   # updater 1 reads queues 6..11 (UPD:187)


def test_philox_known_answers(oracle_lib):
    """Random123 kat_vectors for philox4x32_10."""
    L = oracle_lib
    def ph(ctr, key):
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        L.orc_philox4x32_10(c, k, o)
        return [x for x in o]
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert ph([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_java_util_random_contract(oracle_lib):
    """new Random(42).nextInt() sequence is a published constant of the JDK LCG."""
    L = oracle_lib
    r = JRand()
    L.orc_jrand_seed(C.byref(r), 42)
    got = [L.orc_jrand_next(C.byref(r), 32) for _ in range(3)]
    assert got == [-1170105035, 234785527, -1360544799]
    L.orc_jrand_seed(C.byref(r), 42)
    assert [L.orc_jrand_next_int(C.byref(r), 10) for _ in range(5)] == [0, 3, 8, 4, 0]
    # nextDouble of Random(42): 0.7275636800328681 (== MALLET nextUniform: same next(26)/next(27) recipe)
    L.orc_jrand_seed(C.byref(r), 42)
    assert L.orc_mallet_next_uniform(C.byref(r)) == 0.7275636800328681


def test_java_round(oracle_lib):
    L = oracle_lib
    assert L.orc_java_round(2.5) == 3 and L.orc_java_round(-2.5) == -2
    assert L.orc_java_round(0.49999999999999994) == 0      # the JDK-6 bug value; Java 8 gives 0
    assert L.orc_java_round(310.5) == 311


def test_mallet_next_beta_shapes(oracle_lib):
    L = oracle_lib
    r = JRand()
    L.orc_jrand_seed(C.byref(r), 7)
    # a<1, b=1: Joehnk; Beta(a,1) has mean a/(a+1)
    xs = np.array([L.orc_mallet_next_beta(C.byref(r), 0.31, 1.0) for _ in range(20000)])
    assert np.all((xs >= 0) & (xs <= 1))
    assert abs(xs.mean() - 0.31 / 1.31) < 0.01
    # a>1, b=1: the NaN quirk returns the first in-range N(1, 0.25/(a-1)) proposal -> all values <= 1, clustered near 1
    ys = np.array([L.orc_mallet_next_beta(C.byref(r), 1.1, 1.0) for _ in range(5000)])
    assert np.all((ys >= 0) & (ys <= 1))
    sigma = 0.5 / np.sqrt(0.1)
    # half-normal truncated to [0,1]: mean of 1-|g|*sigma restricted to >=0 is well above Beta(1.1,1)'s 0.52
    assert ys.mean() > 0.45
    # a==b==1 -> plain uniform
    L.orc_jrand_seed(C.byref(r), 42)
    assert L.orc_mallet_next_beta(C.byref(r), 1.0, 1.0) == 0.7275636800328681


def test_kat6_single_view_conditional():
    """M=1: p[0][0]=1, other==0, no nextBeta call; conditional = (1[k in dense] n_dk + gamma alpha_k)(n_wk+beta)/(n_k+betaSum)
    with the token's own count included (Q5) and the indicator only through Q1/Q2."""
    from tests.helpers import small_corpus, make_oracle
    from mvtopicmodel_amd.native import Hyper
    K, V = 5, [11]
    c = small_corpus(K, V, 6, [9], seed=123)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy, init_seed=3)
    z0 = o.get_assignments(0).copy()
    nwk, nk = o.get_counts(0)
    d = 2
    b, e = int(c.doc_off[0][d]), int(c.doc_off[0][d + 1])
    assert e - b >= 2
    res = o.sweep(0, 99, flags=binding.SWEEP_NO_APPLY, trace=[(d, 0, 0)])
    w = int(c.tokens[0][b])
    ndk = np.bincount(z0[b:e], minlength=K).astype(np.float64)
    old = z0[b]
    in_dense = ndk > 0
    ndk[old] -= 1
    if ndk[old] == 0:
        in_dense[old] = False
    phi = (nwk[w] + 0.01) / (nk + 0.01 * V[0])
    want = (in_dense * ndk + 1.0 * 0.1) * phi
    want = want / want.sum()
    got = res["trace"][0]
    assert got[K] == 0.0
    assert np.allclose(got[:K], want, rtol=0, atol=1e-12)


def test_invariants_after_sweeps():
    from tests.helpers import small_corpus, make_oracle
    from mvtopicmodel_amd.native import Hyper
    K, V = 20, [300, 40, 50]
    c = small_corpus(K, V, 64, [30, 4, 6], seed=5)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    for it in range(3):
        st = o.sweep(it, 1234)["stats"]
        assert st["tokens"] == c.total_tokens
        assert st["new_mass_cnt"] == 0
        assert st["topic_doc_mass_cnt"] + st["word_ftree_mass_cnt"] == st["tokens"]
        for m in range(3):
            nwk, nk = o.get_counts(m)
            z = o.get_assignments(m)
            assert nk.sum() == len(z)
            assert np.array_equal(nwk.sum(axis=0), nk)
            tt = np.bincount(c.tokens[m], minlength=V[m])
            assert np.array_equal(nwk.sum(axis=1), tt)
            ref = np.zeros_like(nwk); np.add.at(ref, (c.tokens[m], z), 1)
            assert np.array_equal(ref, nwk)
            hist, dl = o.get_doc_topic_hist(m, 64, 64)
            assert np.array_equal((hist * np.arange(64)[None, :]).sum(axis=1), nk)
            lens = np.diff(c.doc_off[m])
            assert hist.sum(axis=1).tolist() == [int((lens > 0).sum())] * K
            assert np.array_equal(dl, np.bincount(lens[lens > 0], minlength=64))


def test_init_assignments_rule():
    """PTM:499-507: view 0 uniform over K; view m>0 picks among the topics drawn for the entity's view 0."""
    from tests.helpers import small_corpus, make_oracle
    from mvtopicmodel_amd.native import Hyper
    K, V = 50, [200, 30]
    c = small_corpus(K, V, 40, [12, 5], seed=9)
    o = make_oracle(c, Hyper.defaults(K, V), init_seed=1)
    z0, z1 = o.get_assignments(0), o.get_assignments(1)
    assert z0.min() >= 0 and z0.max() < K
    for d in range(c.D):
        a = set(z0[c.doc_off[0][d]:c.doc_off[0][d + 1]].tolist())
        b = set(z1[c.doc_off[1][d]:c.doc_off[1][d + 1]].tolist())
        if a:
            assert b <= a
    # the stream is java.util.Random(1): first draw nextInt(50)
    r = JRand(); L = binding.lib()
    L.orc_jrand_seed(C.byref(r), 1)
    assert z0[0] == L.orc_jrand_next_int(C.byref(r), K)


def test_philox_p_matches_contract():
    """Device contract for the view weights: per-(doc,pair) Philox stream, 3-decimal rounding, symmetric."""
    from tests.helpers import small_corpus, make_oracle
    from mvtopicmodel_amd.native import Hyper
    K, V = 8, [50, 10, 10]
    c = small_corpus(K, V, 30, [10, 3, 3], seed=2)
    o = make_oracle(c, Hyper.defaults(K, V))
    p = o.draw_p_philox(77, 3)
    assert p.shape == (30, 3, 3)
    assert np.all(p[:, [0, 1, 2], [0, 1, 2]] == 1.0)
    assert np.array_equal(p, p.transpose(0, 2, 1))
    assert np.all(np.abs(p * 1000 - np.round(p * 1000)) < 1e-9)
    assert np.array_equal(p, o.draw_p_philox(77, 3))
    assert not np.array_equal(p, o.draw_p_philox(77, 4))
    # shard invariance: entity ids are global
    sub = c.slice_docs(10, 20)
    o2 = make_oracle(sub, Hyper.defaults(K, V))
    assert np.array_equal(o2.draw_p_philox(77, 3, doc_id_base=10), p[10:20])


def _kat7_run_oracle(seed, with_inactive, m_t, pos_t):
    from tests import kat7
    doc_off, toks, z = kat7.corpus()
    hy = kat7.hyper(with_inactive); p = kat7.view_weights(); nwk, nk = kat7.global_counts()
    o = binding.Oracle(kat7.K, kat7.V)
    for m in range(kat7.M):
        o.set_corpus(m, doc_off[m], toks[m]); o.set_assignments(m, z[m])
    o.set_hyper(hy["alpha"], hy["alpha_sum"], hy["beta"], hy["beta_sum"], hy["gamma"],
                np.full((3, 3), 0.31), np.ones((3, 3)), hy["inactive"])
    for m in range(kat7.M):
        o.set_counts(m, nwk[m], nk[m])
    res = o.sweep(0, seed, p=p, flags=binding.SWEEP_NO_APPLY, trace=[(0, m_t, pos_t)])
    za = [o.get_assignments(m) for m in range(kat7.M)]
    want, facts = kat7.expected_conditional(hy, p, nwk, nk, kat7.entity_slices(doc_off, z),
                                            kat7.entity_slices(doc_off, za), m_t, pos_t)
    return res["trace"][0], want, facts


@pytest.mark.parametrize("with_inactive,m_t,pos_t,seed", [(False, 1, 3, 150), (False, 2, 1, 9), (True, 1, 4, 12), (True, 2, 1, 5)])
def test_kat7_multi_view_mid_document_conditional(with_inactive, m_t, pos_t, seed):
    """KAT-7 (tests/kat7.py): the multi-view full conditional at a mid-document token of a later view, after a topic
    has been removed from the dense list (WRK:441-468), a topic outside the list has been entered (Q1) and the removed
    topic re-sampled (Q2) -- derived in numpy from SURVEY §8a step 4 without the oracle; with_inactive adds a non-empty
    inActiveTopicIndex (newTopicMass WRK:413-418,515; zero leaves PTM:2670-2671)."""
    got, want, facts = _kat7_run_oracle(seed, with_inactive, m_t, pos_t)
    assert facts["removed"] and facts["entered_outside_list"] and facts["reentered_removed"]      # the scenario really is the hard one
    assert abs(want.sum() - 1.0) < 1e-12
    assert (want[7] > 0) == with_inactive
    assert np.allclose(got, want, rtol=0, atol=1e-12)


def test_kat7_every_history():
    """The same derivation against 120 different sampled histories (seeds), every later-view position."""
    worst = 0.0
    for seed in range(40):
        for with_inactive in (False, True):
            for (m_t, pos_t) in ((0, 5), (1, 0), (1, 4), (2, 2)):
                got, want, _ = _kat7_run_oracle(seed, with_inactive, m_t, pos_t)
                worst = max(worst, float(np.abs(got - want).max()))
    assert worst < 1e-12


class _PyJavaRandom:
    """java.util.Random, documented contract, in plain Python (independent of the C oracle)."""

    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
        self.have = False
        self.nxt = 0.0

    def next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.s >> (48 - bits)
        return v - (1 << bits) if v >= (1 << (bits - 1)) and bits == 32 else v

    def uniform(self):                       # MALLET Randoms.nextUniform == nextDouble
        return ((self.next(26) << 27) + self.next(27)) / float(1 << 53)

    def gaussian(self):                      # MALLET Randoms.nextGaussian: Box-Muller, caches the sine twin
        import math
        if not self.have:
            v1, v2 = self.uniform(), self.uniform()
            x1 = math.sqrt(-2 * math.log(v1)) * math.cos(2 * math.pi * v2)
            self.nxt = math.sqrt(-2 * math.log(v1)) * math.sin(2 * math.pi * v2)
            self.have = True
            return x1
        self.have = False
        return self.nxt

    def beta(self, a, b):                    # MALLET Randoms.nextBeta as read from the 2.0.8 class file
        import math
        if a == 1 and b == 1:
            return self.uniform()
        if a >= 1 and b >= 1:
            A, B = a - 1, b - 1
            C_ = A + B
            L = C_ * math.log(C_)
            mu, sigma = A / C_, 0.5 / math.sqrt(C_)
            y = self.gaussian(); x = sigma * y + mu
            while x < 0 or x > 1:
                y = self.gaussian(); x = sigma * y + mu
            u = self.uniform()
            while True:
                with np.errstate(all="ignore"):
                    t2 = np.float64(B) * np.log(np.float64(1 - x) / np.float64(B))     # b == 1: 0 * log(./0) = NaN -> comparison false
                    rhs = A * math.log(x / A) + t2 + L + 0.5 * y * y
                if not (math.log(u) >= rhs):
                    return x
                y = self.gaussian(); x = sigma * y + mu
                while x < 0 or x > 1:
                    y = self.gaussian(); x = sigma * y + mu
                u = self.uniform()
        v1, v2 = math.pow(self.uniform(), 1 / a), math.pow(self.uniform(), 1 / b)
        while v1 + v2 > 1:
            v1, v2 = math.pow(self.uniform(), 1 / a), math.pow(self.uniform(), 1 / b)
        return v1 / (v1 + v2)


def test_kat8_view_weights_from_the_mallet_stream_independent_restatement():
    """WRK:327-337 drawn with the worker's MALLET Randoms (java.util.Random LCG -> nextUniform -> nextBeta ->
    Math.round(1000 x)/1000), restated in plain Python from the documented JDK contract and the class-file reading of
    nextBeta -- no call into the C oracle for the expected value -- for the Joehnk branch (a < 1), the Gaussian-proposal
    branch with its b == 1 NaN quirk (a > 1) and the beta == 0.0001 clamp of Q4."""
    import math
    from tests.helpers import small_corpus, make_oracle
    from mvtopicmodel_amd.native import Hyper
    K, V = 6, [40, 9, 9]
    c = small_corpus(K, V, 25, [8, 3, 3], seed=4)
    for pa, clamp in ((0.31, False), (1.1, False), (0.74, True)):
        hy = Hyper.defaults(K, V, p_a=pa)
        if clamp:
            hy.beta[2] = 0.0001; hy.beta_sum[2] = 0.0001 * V[2]
        o = make_oracle(c, hy)
        got = o.draw_p_mallet(12345)
        r = _PyJavaRandom(12345)
        want = np.zeros((c.D, 3, 3))
        for d in range(c.D):
            for m in range(3):
                for j in range(m, 3):
                    if m == j:
                        pr = 1.0
                    else:
                        x = 1000 * r.beta(pa, 1.0)
                        pr = float(math.floor(x + 0.5)) / 1000.0            # Math.round for non-negative finite x
                    want[d, m, j] = 0.0 if (j != 0 and hy.beta[j] == 0.0001) else pr
                    want[d, j, m] = 0.0 if (m != 0 and hy.beta[m] == 0.0001) else pr
        assert np.array_equal(got, want), (pa, clamp)
        if clamp:
            assert np.all(got[:, 2, 2] == 0) and np.all(got[:, 0, 2] == 0) and np.any(got[:, 2, 0] > 0)


def test_numpy_java_random_init_equals_the_host_mirror_and_the_jdk_known_answers():
    """mvtopicmodel_amd/java_init.py (what bench.py, the tools and the full-size tests draw the initial assignments with) against the
    documented java.util.Random answers and against the C++ restatement inside the host mirror, which does it draw by draw:
    bounds that are powers of two, bounds that are not, entities without a text view, and a bound large enough for the rejection
    loop of nextInt to run thousands of times."""
    from mvtopicmodel_amd import java_init
    from hostmirror.binding import init_assignments as ref
    assert java_init.java_next_ints(42, np.full(5, 10)).tolist() == [0, 3, 8, 4, 0]
    rng = np.random.default_rng(0)
    for K in (7, 64, 400):
        D = 1500
        lens = [rng.integers(0, 40, D), rng.integers(0, 5, D), rng.integers(0, 9, D)]
        lens[0][::5] = 0
        offs = [np.concatenate([[0], np.cumsum(l)]).astype(np.int64) for l in lens]
        for a, b in zip(ref(K, offs, seed=1), java_init.init_assignments(K, offs, 1)):
            assert np.array_equal(a, b)
    off = [np.array([0, 200000], dtype=np.int64)]
    big = (1 << 30) + 12345
    assert np.array_equal(ref(big, off, 5)[0], java_init.init_assignments(big, off, 5)[0])


@pytest.mark.parametrize("K,cell16", [(200, 1), (400, 1), (400, 0), (600, 1), (1000, 1), (1000, 0)])
def test_kat9_the_live_rows_tree_branch_samples_the_leaves_of_the_reference(oracle_lib, K, cell16):
    """The tree branch of a live sweep in its live-rows form (orc_row_sample_live = row_sample_live of the HIP kernels, which the GPU tests
    hold equal to it integer for integer) must sample topic k with the probability FTree.sample gives it on a tree of the same leaves
    (FT:111-136 over PTM:2670-2678): leaf_k / sum_j leaf_j with leaf_k = coef_k * (n_wk + beta).  Derived here without the function under
    test: the leaves in fp64 from the definition; the function is swept over a fine grid of u2 and the measure of the u2 it maps to each
    topic compared with the leaf's share.  K = 600 and 1000 on the mirror are rows of two register batches (the stored first-batch mass
    decides where the search starts); K = 1000 on the 32-bit table is four batches scanned in order."""
    L = oracle_lib
    rng = np.random.default_rng(1000 + K + cell16)
    beta = 0.01
    n = np.zeros(K, dtype=np.int32)
    nz = rng.choice(K, size=max(8, K // 9), replace=False)                 # a sparse row, like a word's
    n[nz] = rng.integers(1, 400, size=len(nz))
    n[nz[0]] = 5000                                                         # and one dominant topic
    coef = (1.0 / (rng.integers(50, 50000, size=K) + 50.0)).astype(np.float32)
    coef[rng.choice(K, size=K // 10, replace=False)] = 0.0                  # inactive topics: no leaf
    smp = np.cumsum((coef * np.float32(beta)).astype(np.float32), dtype=np.float32).astype(np.float32)
    # (the running sums the library keeps are a sequential fp32 sum: restate that exactly)
    run = np.float32(0.0)
    for k in range(K):
        run = np.float32(run + np.float32(coef[k] * np.float32(beta)))
        smp[k] = run
    S = float(smp[K - 1])
    leaves = coef.astype(np.float64) * (n.astype(np.float64) + np.float64(np.float32(beta)))
    root = float(S + np.dot(coef.astype(np.float64), n.astype(np.float64)))
    b0 = 512 if cell16 else 256
    mass0 = np.float32(np.dot(coef[:b0].astype(np.float64), n[:b0].astype(np.float64)))
    G = 400000
    hits = np.zeros(K, dtype=np.int64)
    rowp, cfp, smpp = n.ctypes.data, coef.ctypes.data, smp.ctypes.data
    for i in range(G):
        u2 = (i + 0.5) / G
        k = L.orc_row_sample_live(rowp, cfp, smpp, K, cell16, C.c_float(u2), C.c_float(root), C.c_float(float(mass0)))
        assert 0 <= k < K
        hits[k] += 1
    got = hits / G
    want = leaves / leaves.sum()
    # fp32 arithmetic (2^-24 relative per operation over at most K terms), the grid (1 / G per interval end) and the order of the
    # cells inside a lane (even cells first) move interval ends, not masses: 2e-5 absolute per topic is an order above all of them
    assert np.abs(got - want).max() < 2e-5, (np.abs(got - want).argmax(), np.abs(got - want).max())
    assert hits[coef == 0].sum() == 0                                       # an inactive topic is never drawn
    assert abs(leaves.sum() - root) < 1e-6 * root
