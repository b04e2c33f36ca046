"""Host-side samplers of optimizeDP (PTM:2440-2591) and optimizeGamma (PTM:2369-2438): the C++ host
mirror's restatement (hostmirror/knowceans_samplers.h, java_random.h) against the pure-Python oracle
(oracle/dp_samplers.py), plus anchors that do not depend on either: numpy's MT19937 for the Cokus
stream, exact Antoniak probabilities, and Gamma/Beta moments."""
import math

import numpy as np
import pytest

from oracle import dp_samplers as orc


def test_cokus_stream_is_mt19937_with_the_1998_seeding():
    from hostmirror.binding import cokus_stream
    n = 2000                                                       # crosses three state reloads
    got = cokus_stream(n)
    # independent generator: numpy's MT19937 primed with the 69069-LCG state of seed 4357
    key = np.empty(624, dtype=np.uint32)
    x = 4357 | 1
    for i in range(624):
        key[i] = x
        x = (x * 69069) & 0xFFFFFFFF
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": 624}}
    want = bg.random_raw(n).astype(np.uint32)
    assert np.array_equal(got, want)
    c = orc.Cokus()
    assert [c.rand() for _ in range(n)] == want.tolist()


def test_stirling_rows_and_first_antoniak_call_follow_the_exact_law():
    # unsigned Stirling numbers of the first kind, row n = 5: 24 50 35 10 1 (normalised by the max)
    s = orc.StaticSamplers()
    row = s.stirling(5)
    assert np.allclose(np.array(row) * 50, [24, 50, 35, 10, 1])
    assert abs(s.logmaxss[4] - math.log(50)) < 1e-12
    # P(m tables | alpha, n) ∝ s(n, m) alpha^m : the first call on a fresh cache samples exactly that
    alpha, n = 0.7, 6
    st = [120, 274, 225, 85, 15, 1]
    w = np.array([st[m] * alpha ** m for m in range(n)]); cdf = np.cumsum(w) / w.sum()
    u = orc.Cokus().randDouble()
    want = int(np.searchsorted(cdf, u, side="left")) + 1
    assert orc.StaticSamplers().rand_antoniak(alpha, n) == want


def test_rand_antoniak_sequences_match_including_the_cache_corruption_quirk():
    from hostmirror.binding import rand_antoniak_seq
    rng = np.random.RandomState(4)
    n = rng.randint(2, 60, 400).astype(np.int32)
    n[::50] = 200                                                   # extends the cache from already modified rows
    alpha = rng.gamma(1.0, 1.0, 400)
    got = rand_antoniak_seq(alpha, n)
    s = orc.StaticSamplers()
    want = [s.rand_antoniak(float(a), int(k)) for a, k in zip(alpha, n)]
    assert got.tolist() == want
    assert (got >= 1).all() and (got <= n + 1).all()
    # the quirk: the cached row is scaled and prefix-summed in place, so a repeated call does not see s(n, .)
    s2 = orc.StaticSamplers()
    fresh = list(s2.stirling(8)); s2.rand_antoniak(2.0, 8)
    assert s2.stirling(8) != fresh
    assert all(b >= a for a, b in zip(s2.stirling(8), s2.stirling(8)[1:]))   # now a running sum
    # beyond MAXSTIRLING the Java code throws (ArrayIndexOutOfBounds) and optimizeDP falls back to 1 table
    assert rand_antoniak_seq([1.0, 1.0], [20001, 3]).tolist()[0] == -1
    with pytest.raises(IndexError):
        orc.StaticSamplers().rand_antoniak(1.0, 20001)


@pytest.mark.parametrize("kind,a,b", [("gamma", 0.3, 0), ("gamma", 1.0, 0), ("gamma", 7.5, 0), ("gamma_scale", 2.5, 0.25),
                                      ("beta", 2.0, 5.0), ("beta", 11.0, 0.0), ("bernoulli", 0.3, 0)])
def test_random_samplers_streams_match_oracle_and_moments(kind, a, b):
    from hostmirror.binding import random_samplers_stream
    n = 4000
    got = random_samplers_stream(12345, kind, a, b, n)
    samp = orc.RandomSamplers(orc.JavaRandom(12345))
    f = {"gamma": lambda: samp.rand_gamma(a), "gamma_scale": lambda: samp.rand_gamma(a, b),
         "beta": lambda: samp.rand_beta(a, b), "bernoulli": lambda: float(samp.rand_bernoulli(a))}[kind]
    want = np.array([f() for _ in range(n)])
    assert np.array_equal(got, want)
    mean = {"gamma": a, "gamma_scale": a * b, "beta": a / (a + b), "bernoulli": a}[kind]
    var = {"gamma": a, "gamma_scale": a * b * b, "beta": a * b / ((a + b) ** 2 * (a + b + 1)), "bernoulli": a * (1 - a)}[kind]
    assert abs(got.mean() - mean) < 5 * math.sqrt(var / n) + 1e-12


@pytest.mark.parametrize("alpha,beta", [(0.05, 1.0), (0.5, 1.0), (1.0, 2.0), (3.7, 1.0), (250.0, 1.0)])
def test_mallet_next_gamma_matches_oracle_and_moments(alpha, beta):
    from hostmirror.binding import mallet_next_gamma_stream
    n = 4000
    got = mallet_next_gamma_stream(99, alpha, beta, n)
    r = orc.JavaRandom(99)
    want = np.array([orc.mallet_next_gamma(r, alpha, beta) for _ in range(n)])
    assert np.array_equal(got, want)
    assert abs(got.mean() - alpha * beta) < 5 * math.sqrt(alpha * beta * beta / n)
    with pytest.raises(ValueError):
        mallet_next_gamma_stream(1, 0.0, 1.0, 1)


def test_optimize_dp_and_gamma_oracle_hand_case():
    """Two topics, one view.  Topic 1 has no tokens: it stays in the inactive set and gets the 0.0001
    pseudo-draw of sampleDirichlet; single-token entities contribute one table each without a draw."""
    K, M = 2, 1
    st = orc.DPState(K, M, [[0.1, 0.1, 0.1]], [1.0])
    tdc = [[[0, 3, 0, 0], [4, 0, 0, 0]]]                            # topic 0: three entities with one token
    statics = orc.StaticSamplers()
    orc.optimize_dp(st, tdc, statics, orc.JavaRandom(5))
    assert st.inactive == {1}
    assert statics.maxnn == 3                                       # only the root-level call (n = ceil(3)) touched the cache
    assert st.rootTablesCnt > 10 and st.rootTablesCnt <= 13        # gammaRoot + 1..3 root tables
    assert abs(sum(st.alpha[0]) - 1) < 1e-12 and abs(st.alphaSum[0] - 1) < 1e-12
    assert st.alpha[0][1] < 1e-3 < st.alpha[0][0]
    g0 = st.gamma[0]
    orc.optimize_gamma(st, [[0, 2, 1, 0]], orc.RandomSamplers(orc.JavaRandom(6)))
    assert st.gamma[0] > 0 and st.gamma[0] != g0 and st.gammaRoot > 0 and st.gammaView[0] > 0
