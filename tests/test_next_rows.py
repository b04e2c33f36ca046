"""SURVEY §8f "next" rows: the statistics the host's optimizeBeta / optimizeP read, and modelLogLikelihood."""
import math

import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper
from tests.helpers import make_native, make_oracle, small_corpus


# ------------------------------ CPU: oracle pins ------------------------------
def test_log_gamma_stirling_tracks_lgamma(oracle_lib):
    for z in (1e-4, 0.01, 0.1, 0.5, 1.0, 1.9999, 2.0, 3.7, 40.25, 1e3, 1e6):
        got = oracle_lib.orc_log_gamma_stirling(z)
        assert abs(got - math.lgamma(z)) < 1e-5 * max(1.0, abs(math.lgamma(z))), z    # truncation error 1/(1680 z^7) <= 4.7e-6 at z=2
    # logGammaStirling(1) = shift once: stirling(2) - log(1)
    z = 2.0
    st2 = math.log(6.283185307179586) / 2.0 + (z - 0.5) * math.log(z) - z + 1 / (12.0 * z) - 1 / (360.0 * z * z * z) + 1 / (1260.0 * z ** 5)
    assert oracle_lib.orc_log_gamma_stirling(1.0) == st2 - math.log(1.0)


def test_mallet_digamma_as_compiled(oracle_lib):
    # the 2.0.8 class file has the series coefficients as integer quotients (= 0): psi(z) = log z - 1/(2z) for z >= 9.5
    assert oracle_lib.orc_mallet_digamma(50.0) == math.log(50.0) - 0.5 * (1 / 50.0)
    z, psi = 3.0, 0.0
    while z < 9.5:
        psi -= 1 / z; z += 1
    assert oracle_lib.orc_mallet_digamma(3.0) == psi + (math.log(z) - 0.5 * (1 / z))
    assert oracle_lib.orc_mallet_digamma(1e-7) == -0.5772156649015329 - 1 / 1e-7


def test_learn_symmetric_concentration_fixed_point(oracle_lib):
    """Known structure: observations drawn from a symmetric Dirichlet-multinomial; the estimate must
    move from the starting value towards a finite positive concentration and be reproducible."""
    rng = np.random.RandomState(0)
    V, K = 200, 30
    p = rng.dirichlet(np.full(V, 0.05), K)
    counts = np.stack([rng.multinomial(400, p[k]) for k in range(K)])      # [K][V]
    ch = np.bincount(counts[counts > 0], minlength=counts.max() + 1).astype(np.int32); ch[0] = 0
    lens = np.bincount(counts.sum(axis=1)).astype(np.int32)
    a = oracle_lib.orc_learn_symmetric_concentration(ch.ctypes.data, len(ch), lens.ctypes.data, len(lens), V, 0.01 * V)
    b = oracle_lib.orc_learn_symmetric_concentration(ch.ctypes.data, len(ch), lens.ctypes.data, len(lens), V, 0.01 * V)
    assert a == b and 0.5 < a < 100          # true concentration 0.05*200 = 10; MALLET's biased digamma lands in the right decade
    assert abs(a - 10) < 6


def test_optimize_p_sums_hand_case():
    """Two views; entity 0: view1 shorter, 2 of its 3 tokens share a topic with view 0 -> 2/3; entity 1: equal lengths,
    the TreeMap keeps only the later view -> no statistic; entity 2: view 1 absent."""
    from oracle.binding import Oracle
    K, V = 6, [10, 10]
    off0 = np.array([0, 4, 6, 9], dtype=np.int64); off1 = np.array([0, 3, 5, 5], dtype=np.int64)
    o = Oracle(K, V)
    o.set_corpus(0, off0, np.zeros(9, dtype=np.int32)); o.set_corpus(1, off1, np.zeros(5, dtype=np.int32))
    o.set_assignments(0, np.array([1, 1, 2, 3, 0, 0, 4, 4, 5], dtype=np.int32))
    o.set_assignments(1, np.array([1, 5, 3, 0, 0], dtype=np.int32))
    s = o.optimize_p_sums()
    x = 1.0 / 3.0
    assert s[1, 0] == s[0, 1] == (0.0 + x) + x
    assert s[0, 0] == 0 and s[1, 1] == 0


# ------------------------------ GPU: parity ------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("K,V,D,lam", [(20, [300, 40, 50], 90, [30, 4, 6]), (100, [900], 60, [70]), (400, [3000, 300], 50, [120, 8])])
def test_statistics_and_log_likelihood_match_oracle(K, V, D, lam):
    c = small_corpus(K, V, D, lam, 500 + K)
    hy = Hyper.defaults(K, V)
    hy.alpha[:] = np.linspace(0.02, 0.3, K + 1)[None, :]
    hy.alpha_sum[:] = hy.alpha[:, :K].sum(axis=1)
    hy.gamma[:] = np.linspace(0.8, 1.3, c.M)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    for it in range(3):
        # optimizeBeta's countHistogram: exact integers
        for m in range(c.M):
            mx = int(max(o.get_counts(m)[0].max(), 1))
            assert np.array_equal(o.count_histogram(m, mx + 1), s.get_count_histogram(m, mx + 1))
        # optimizeP's per-pair sums: same per-entity arithmetic, summed in entity order -> bit-identical
        assert np.array_equal(o.optimize_p_sums(), s.view_overlap_sums())
        # modelLogLikelihood: device log() and a different summation order -> relative 1e-12
        lo, ls = o.model_log_likelihood(), s.model_log_likelihood()
        assert np.all(np.isfinite(ls)) and np.all(ls < 0)
        assert np.allclose(lo, ls, rtol=1e-12, atol=0), (lo, ls)
        o.sweep(it, 31); s.sweep(it, 31)
    s.close()


@pytest.mark.gpu
def test_log_likelihood_rises_over_sweeps():
    """The only convergence signal the reference logs (PTM:1302-1304): LL/token must improve from the random start."""
    K, V = 50, [2000, 200]
    c = small_corpus(K, V, 400, [80, 8], 77)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    ll0 = s.model_log_likelihood()
    for it in range(15):
        s.sweep(it, 5)
    ll1 = s.model_log_likelihood()
    assert np.all(ll1 > ll0)
    s.close()


@pytest.mark.gpu
def test_inferencer_frozen_model_mode():
    """the inferencer is a second caller of the same worker with nut=0 (INF:211-294):
    trees without gamma*alpha (INF:557-586), initial topics drawn from the trees (INF:169-199), the
    model never updated, out-of-vocabulary tokens skipped (WRK:427) but counted as topic 0."""
    from mvtopicmodel_amd import NativeSampler
    from mvtopicmodel_amd.native import SWEEP_FROZEN
    from oracle.binding import Oracle, SWEEP_FROZEN as ORC_FROZEN
    K, V = 40, [600, 70]
    train = small_corpus(K, V, 200, [50, 6], 901)
    hy = Hyper.defaults(K, V)
    o_tr = make_oracle(train, hy)
    for it in range(3):
        o_tr.sweep(it, 1)
    model = [o_tr.get_counts(m) for m in range(2)]

    new = small_corpus(K, V, 60, [40, 5], 902)
    tok0 = new.tokens[0].copy(); tok0[::17] = V[0] + 3          # out-of-vocabulary types (new alphabet entries)
    from mvtopicmodel_amd.synth import Corpus
    new = Corpus(K, V, new.doc_off, [tok0, new.tokens[1]])
    hyi = Hyper.defaults(K, V, p_a=0.2)                         # INF: p_a = 0.2, p_b = 1
    o = Oracle(K, V); s = NativeSampler(K, V)
    for be in (o, s):
        for m in range(2):
            be.set_corpus(m, new.doc_off[m], new.tokens[m])
            be.set_counts(m, *model[m])
    o.set_hyper(hyi.alpha, hyi.alpha_sum, hyi.beta, hyi.beta_sum, hyi.gamma, hyi.p_a, hyi.p_b, None)
    s.set_hyper(hyi)
    o.build_inference_trees(); s.build_inference_trees()
    for m, w in [(0, 0), (0, 599), (1, 33)]:
        assert np.array_equal(o.get_tree(m, w), s.get_tree(m, w))
    o.init_assignments_from_trees(77); s.init_assignments_from_trees(77)
    for m in range(2):
        assert np.array_equal(o.get_assignments(m), s.get_assignments(m))
    assert (s.get_assignments(0)[::17] == 0).all()
    for it in range(1, 4):                                      # numIterations = 10 in the reference; 3 suffice here
        ro = o.sweep(it, 77, flags=ORC_FROZEN)
        rs = s.sweep(it, 77, flags=SWEEP_FROZEN)
        assert rs.changed == 0 == ro["stats"]["changed"]
        assert rs.oov_skipped == ro["stats"]["oov_skipped"] > 0
        for m in range(2):
            assert np.array_equal(o.get_assignments(m), s.get_assignments(m))
            a, b = s.get_counts(m)
            assert np.array_equal(a, model[m][0]) and np.array_equal(b, model[m][1])      # frozen model
    assert (s.get_assignments(0)[::17] == 0).all()              # OOV tokens are never resampled
    s.close()


@pytest.mark.gpu
def test_doc_topic_proportions_and_print_document_topics(tmp_path):
    """printDocumentTopics PTM:2820-2960 (text half): device proportions bit-exact against the numpy restatement, and the
    host mirror's text -- descending weight, ties by descending topic id (IDSorter.compareTo), cut at threshold/max, the growing line printed once per
    retained topic -- against the same text assembled in Python."""
    from hostmirror.binding import FastQMVWVParallelTopicModel, java_double_to_string
    from mvtopicmodel_amd import synth
    from oracle import doc_topics as dto
    K, V = 12, [150, 30]
    c = synth.generate(K, V, 40, [25, 4], seed=77, chunk_docs=4096)
    training = [(np.arange(c.D, dtype=np.int64) + 1000, c.doc_off[m], c.tokens[m], V[m]) for m in range(2)]
    model = FastQMVWVParallelTopicModel(K, 2, 0.1, 0.01)
    model.setNumIterations(3); model.setBurninPeriod(200); model.setOptimizeInterval(50); model.setRandomSeed(11)
    model.addInstances(training)
    model.estimate()
    z = [model.get_view(m)[3] for m in range(2)]
    hy = Hyper.defaults(K, V)
    pmean = np.array([[1.0, 0.4], [0.4, 1.0]]); discr = np.array([1.0, 0.7])
    w = [1.0 * pmean[0, 0], discr[1] * pmean[0, 1]]
    want = dto.doc_topic_proportions(K, c.doc_off, z, hy.alpha, hy.alpha_sum, hy.gamma, w)
    assert np.allclose(want.sum(axis=1), 1.0)

    from mvtopicmodel_amd import NativeSampler
    s = NativeSampler(K, V)
    for m in range(2):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z[m])
    s.set_hyper(hy)
    got = s.doc_topic_proportions(w)
    assert np.array_equal(got, want)
    assert np.array_equal(s.doc_topic_proportions(w, 7, 19), want[7:19])
    s.close()

    # threshold 0, every topic: the many topics an entity does not hold have equal weights and come out by descending id
    f0 = tmp_path / "doc_topics_all.txt"
    model.printDocumentTopics(f0, 0.0, -1, discr_weight=discr, p_mean=pmean)
    assert f0.read_text() == dto.print_document_topics(want, [str(1000 + d) for d in range(c.D)], 0.0, -1, java_double_to_string)
    f = tmp_path / "doc_topics.txt"
    model.printDocumentTopics(f, 0.05, 4, discr_weight=discr, p_mean=pmean)
    text = f.read_text()
    assert text == dto.print_document_topics(want, [str(1000 + d) for d in range(c.D)], 0.05, 4, java_double_to_string)
    lines = text.splitlines()
    assert lines[0] == "#doc name topic proportion ..." and lines[1].startswith("0\t1000\t")
    assert lines[2].startswith(lines[1])                                # the builder keeps growing within an entity
    model.close()


@pytest.mark.gpu
def test_gamma_doc_statistics_distribution_and_determinism():
    """optimizeGamma's document-level sums (PTM:2415-2433) from the device: qs = sum Bernoulli(j/(j+gamma)),
    qw = sum log Beta(gamma+1, j) over the entities with the view.  The reference's stream for them cannot be seeded, so the
    bar is the distribution: E qs = sum j/(j+gamma), Var qs = sum p(1-p); E qw = sum psi(gamma+1) - psi(gamma+1+j),
    Var qw = sum psi'(gamma+1) - psi'(gamma+1+j); plus determinism for a seed and independence of document shards."""
    from scipy.special import digamma, polygamma
    from mvtopicmodel_amd import NativeSampler, synth
    K, V = 10, [300, 40]
    c = synth.generate(K, V, 40000, [30, 4], seed=41, chunk_docs=8192)
    s = NativeSampler(K, V)
    for m in range(2):
        s.set_corpus(m, c.doc_off[m], c.tokens[m])
    for m, g in ((0, 1.0), (1, 0.37), (0, 6.5)):
        j = np.diff(c.doc_off[m]).astype(np.float64); j = j[j > 0]
        p = j / (j + g)
        e_qs, v_qs = p.sum(), (p * (1 - p)).sum()
        e_qw = (digamma(g + 1) - digamma(g + 1 + j)).sum()
        v_qw = (polygamma(1, g + 1) - polygamma(1, g + 1 + j)).sum()
        draws = np.array([s.gamma_doc_statistics(m, g, 777, r) for r in range(24)])
        assert abs(draws[:, 0].mean() - e_qs) < 4 * np.sqrt(v_qs / 24) and abs(draws[:, 1].mean() - e_qw) < 4 * np.sqrt(v_qw / 24)
        assert 0.4 * v_qs < draws[:, 0].var(ddof=1) < 2.2 * v_qs and 0.4 * v_qw < draws[:, 1].var(ddof=1) < 2.2 * v_qw
        assert s.gamma_doc_statistics(m, g, 777, 3) == tuple(draws[3])                       # deterministic for (seed, round)
        assert s.gamma_doc_statistics(m, g, 778, 3) != tuple(draws[3])
    # two shards with global entity ids: the per-entity draws are the same, so the sums agree up to the summation order
    lo = 17000
    a = NativeSampler(K, V); b = NativeSampler(K, V, doc_id_base=lo)
    sa, sb = c.slice_docs(0, lo), c.slice_docs(lo, c.D)
    for m in range(2):
        a.set_corpus(m, sa.doc_off[m], sa.tokens[m]); b.set_corpus(m, sb.doc_off[m], sb.tokens[m])
    whole = s.gamma_doc_statistics(0, 1.0, 777, 5)
    pa, pb = a.gamma_doc_statistics(0, 1.0, 777, 5), b.gamma_doc_statistics(0, 1.0, 777, 5)
    assert pa[0] + pb[0] == whole[0] and abs(pa[1] + pb[1] - whole[1]) < 1e-9 * abs(whole[1])
    a.close(); b.close(); s.close()


@pytest.mark.gpu
def test_dp_table_statistics_distribution_and_determinism():
    """optimizeDP's view-table simulation (PTM:2454-2488) from the device: a cell (topic t, count i > 1) that holds n entities adds
    n times ONE draw of the number of tables a CRP(conc_t) makes of i items; a cell with i == 1 adds its entities; a topic is active iff
    a cell with i >= 1 holds an entity.  The reference's stream (ThreadLocalRandom) cannot be seeded and its Stirling-table sampler
    scales the cached row in place, so the bar is the Antoniak distribution itself: E = sum_l a/(a+l), Var = sum_l a l/(a+l)^2, also
    against the Stirling-number form of the oracle's restatement; plus determinism for (seed, round)."""
    from mvtopicmodel_amd import NativeSampler
    from oracle import dp_samplers
    K, V = 2048, [10]
    s = NativeSampler(K, V)
    L = 64
    for a, i in ((0.1, 50), (1.7, 12), (25.0, 63), (0.003, 2)):
        hist = np.zeros((K, L), dtype=np.int32)
        hist[:, i] = 3                                          # three entities in the cell: mk = 3 * tables
        hist[5, 1] = 7                                          # ... and seven with a single token of topic 5
        hist[9, :] = 0                                          # topic 9 holds nothing: inactive
        conc = np.full(K, a)
        l = np.arange(i, dtype=np.float64)
        e, v = (a / (a + l)).sum(), (a * l / (a + l) ** 2).sum()
        draws = []
        for r in range(6):
            mk, act = s.dp_table_statistics(0, hist, conc, 99, r)
            assert act[9] == 0 and mk[9] == 0 and act.sum() == K - 1
            t = mk.copy(); t[5] -= 7; t = np.delete(t, 9) / 3.0
            assert np.all(t == np.round(t)) and t.min() >= 1 and t.max() <= i
            draws.append(t)
        d = np.concatenate(draws)
        assert abs(d.mean() - e) < 5 * np.sqrt(v / d.size) + 1e-12, (a, i, d.mean(), e)
        assert 0.85 * v - 1e-9 <= d.var() <= 1.15 * v + 1e-9, (a, i, d.var(), v)
        # the Stirling-number form the reference draws from (unscaled row): the same probabilities
        row = np.array(dp_samplers.StaticSamplers().stirling(i), dtype=np.float64)      # a fresh cache: the row as Samplers.java:1052-1084 computes it, not yet scaled
        pmf = row * a ** np.arange(len(row)); pmf /= pmf.sum()
        got = np.bincount(d.astype(int), minlength=i + 1)[1:] / d.size
        assert np.abs(got - pmf).max() < 5 * np.sqrt(0.25 / d.size)
        mk1, _ = s.dp_table_statistics(0, hist, conc, 99, 2)
        assert np.array_equal(np.delete(mk1, 9)[:5] / 3.0, draws[2][:5])                         # deterministic for (seed, round)
        assert a < 0.01 or not np.array_equal(s.dp_table_statistics(0, hist, conc, 100, 2)[0], mk1)
    # the root level's draws (PTM:2491-2517): the same distribution for counts up to MAXSTIRLING, one table beyond (the reference's fallback)
    items = np.array([0, 1, 2, 500, 20000, 20001, 3_000_000] + [800] * 4000, dtype=np.int32)
    g = 1.3
    tb = s.antoniak_draws(items, np.full(len(items), g), 5, 1)
    assert list(tb[:2]) == [0, 1] and tb[2] in (1, 2) and 1 <= tb[3] <= 500 and tb[4] > 1 and list(tb[5:7]) == [1, 1]
    l = np.arange(800, dtype=np.float64)
    e, v = (g / (g + l)).sum(), (g * l / (g + l) ** 2).sum()
    assert abs(tb[7:].mean() - e) < 5 * np.sqrt(v / 4000) and 0.85 * v < tb[7:].var() < 1.15 * v
    assert np.array_equal(tb, s.antoniak_draws(items, np.full(len(items), g), 5, 1))
    s.close()


@pytest.mark.gpu
def test_view_present_but_empty(oracle_lib):
    """An instance with an empty FeatureSequence is not a missing view (MTA:19): `mvhdp_set_view_presence` lets the statistics tell
    them apart (VERDICT r2 missing #5, advisor).  Derived here from the reference's formulas, no oracle run involved:
      * modelLogLikelihood (PTM:3348-3373): the entity counts in modalityCnt and its empty LabelSequence has a backing array of 2
        -> two phantom tokens of topic 0: + lgs(g*a_0 + 2) - lgs(g*a_0) - lgs(g*alphaSum + 2) + lgs(g*alphaSum);
      * totalDocsPerModality / docLengthCounts[0] / bucket 0 of topicDocCounts (PTM:620-651) count it;
      * printDocumentTopics (PTM:2873-2886) refreshes its counts to zeros instead of carrying the previous holder's over."""
    lgs = oracle_lib.orc_log_gamma_stirling
    K, V = 12, [60, 15]
    c = small_corpus(K, V, 30, [12, 3], 55)
    lens1 = np.diff(c.doc_off[1])
    empty = np.flatnonzero(lens1 == 0)
    assert len(empty) >= 3                                         # the generator leaves some entities without the side view
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    ll0 = s.model_log_likelihood()
    hist0, dl0 = s.get_doc_topic_hist(1, 8, 8)
    w = np.array([1.0, 0.7])
    prop0 = s.doc_topic_proportions(w)
    present = (lens1 > 0).astype(np.uint8)
    pe = empty[1:3]                                                # two of them are "present but empty", the rest stay missing
    present[pe] = 1
    s.set_view_presence(1, present)
    ll1 = s.model_log_likelihood()
    ga, gas = hy.gamma[1] * hy.alpha[1][0], hy.gamma[1] * hy.alpha_sum[1]
    extra = len(pe) * (lgs(ga + 2) - lgs(ga) - lgs(gas + 2) + lgs(gas))
    assert ll1[0] == ll0[0] and abs(ll1[1] - (ll0[1] + extra)) < 1e-9 * abs(ll0[1])
    hist1, dl1 = s.get_doc_topic_hist(1, 8, 8)
    assert dl1[0] == dl0[0] + len(pe) and np.array_equal(dl1[1:], dl0[1:])
    assert np.array_equal(hist1[:, 0], hist0[:, 0] + len(pe)) and np.array_equal(hist1[:, 1:], hist0[:, 1:])
    prop1 = s.doc_topic_proportions(w)
    a1 = hy.gamma[1] * hy.alpha[1][:K] / (hy.gamma[1] * hy.alpha_sum[1])          # the view's share with zero counts and length 0
    for d in range(c.D):
        if d in pe:
            z0 = s.get_assignments(0)[c.doc_off[0][d]:c.doc_off[0][d + 1]]
            n0 = np.bincount(z0, minlength=K)
            a0 = (n0 + hy.gamma[0] * hy.alpha[0][:K]) / (len(z0) + hy.gamma[0] * hy.alpha_sum[0])
            assert np.allclose(prop1[d], (w[0] * a0 + w[1] * a1) / w.sum(), rtol=1e-12, atol=0)
    # entities after a present-empty one that lack the view now carry ITS zeros, not an earlier holder's counts
    later_missing = [d for d in range(int(pe[0]) + 1, c.D) if present[d] == 0]
    if later_missing and later_missing[0] < (pe[1] if len(pe) > 1 else c.D):
        assert not np.array_equal(prop1[later_missing[0]], prop0[later_missing[0]])
    with pytest.raises(Exception):
        bad = present.copy(); bad[np.flatnonzero(lens1 > 0)[0]] = 0
        s.set_view_presence(1, bad)                                # an entity with tokens cannot be absent
    s.set_view_presence(1, None)
    assert np.array_equal(s.model_log_likelihood(), ll0)
    s.close()
