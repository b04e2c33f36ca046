"""The C++ host-side mirror of FastQMVWVParallelTopicModel (hostmirror/), driven the way a MALLET
client drives the reference: new ...(K, M, alpha, beta); set*; addInstances(InstanceList[]); estimate()."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper
from tests.helpers import small_corpus

pytestmark = pytest.mark.gpu


def _instance_lists(rng, V):
    """Three views whose instances are NOT aligned: names decide the entity (PTM:437-455)."""
    def mk(names, lens, v):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return (np.array(names, dtype=np.int64), off, rng.randint(0, v, off[-1]).astype(np.int32), v)
    v0 = mk([10, 11, 12, 13, 14, 15], [30, 1, 45, 9, 0, 22], V[0])
    v1 = mk([12, 99, 10, 15], [4, 6, 3, 5], V[1])          # 99 has no text view -> appended as a new entity
    v2 = mk([99, 13, 77], [2, 8, 1], V[2])                   # 77 appended too
    return [v0, v1, v2]


def test_add_instances_and_estimate_match_the_oracle():
    from hostmirror.binding import FastQMVWVParallelTopicModel
    from oracle.binding import Oracle
    K, V = 25, [300, 40, 30]
    rng = np.random.RandomState(3)
    training = _instance_lists(rng, V)
    model = FastQMVWVParallelTopicModel(K, 3, 0.1, 0.01)
    model.setNumIterations(4); model.setBurninPeriod(200); model.setOptimizeInterval(50); model.setRandomSeed(7)
    model.addInstances(training)

    # entity order: view-0 instances in list order, then unmatched instances of later views (PTM:443-455)
    ids0, off0, tok0, z0 = model.get_view(0)
    assert ids0.tolist() == [10, 11, 12, 13, 14, 15, 99, 77]
    assert np.diff(off0).tolist() == [30, 1, 45, 9, 0, 22, 0, 0]
    ids1, off1, tok1, z1 = model.get_view(1)
    assert np.diff(off1).tolist() == [3, 0, 4, 0, 0, 5, 6, 0]
    ids2, off2, tok2, z2 = model.get_view(2)
    assert np.diff(off2).tolist() == [0, 0, 0, 8, 0, 0, 2, 1]
    # tokens landed with their entity
    assert np.array_equal(tok1[off1[2]:off1[3]], training[1][2][0:4])      # name 12 is instance 0 of view 1

    # oracle on the same flattened corpus: same initial draw (java.util.Random(7)) and same counts
    o = Oracle(K, V)
    views = [(off0, tok0), (off1, tok1), (off2, tok2)]
    for m, (off, tok) in enumerate(views):
        o.set_corpus(m, off, tok)
    hy = Hyper.defaults(K, V, p_a=0.2)                       # PTM:1055-1058
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
    o.init_assignments(7)
    for m, z in enumerate((z0, z1, z2)):
        assert np.array_equal(o.get_assignments(m), z), f"initial assignments differ in view {m}"
    o.build_counts()
    for m in range(3):
        a, b = model.get_counts(m)
        assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])

    model.estimate()
    for it in range(1, 5):                                   # PTM:1146,1166-1171
        hy.p_a[:] = min(it / 100 + 0.3, 1.1)
        o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
        o.sweep(it, 7)
    for m in range(3):
        assert np.array_equal(model.get_view(m)[3], o.get_assignments(m)), f"assignments differ in view {m}"
        a, b = model.get_counts(m)
        assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])
    log = model.iteration_log()
    assert len(log) == 4
    total = int(off0[-1] + off1[-1] + off2[-1])
    for ms, st in log:
        assert st["tokens"] == total
        assert st["new_mass_cnt"] + st["topic_doc_mass_cnt"] + st["word_ftree_mass_cnt"] == total
    model.close()


def test_single_view_is_plain_lda_path():
    """BASELINE config 1/2 shape: M=1 (no view weights are drawn, p[0][0]=1, KAT-6)."""
    from hostmirror.binding import FastQMVWVParallelTopicModel
    from mvtopicmodel_amd.java_init import init_assignments
    K, V = 20, [500]
    rng = np.random.RandomState(4)
    lens = rng.poisson(20, 80) + 1
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tok = rng.randint(0, 500, off[-1]).astype(np.int32)
    model = FastQMVWVParallelTopicModel(K, 1, 0.1, 0.01)
    model.setNumIterations(3); model.setRandomSeed(11)
    model.addInstances([(np.arange(80, dtype=np.int64), off, tok, 500)])
    z_init = model.get_view(0)[3].copy()
    assert np.array_equal(z_init, init_assignments(K, [off], 11)[0])
    model.estimate()
    nwk, nk = model.get_counts(0)
    z = model.get_view(0)[3]
    ref = np.zeros_like(nwk); np.add.at(ref, (tok, z), 1)
    assert np.array_equal(ref, nwk) and np.array_equal(nwk.sum(axis=0), nk)
    assert (z != z_init).mean() > 0.3
    model.close()


def _run_schedule(K, V, D, lam, cseed, seed, iters, burnin, interval, alpha=0.1, shards=1):
    """estimate() of the host mirror against the same schedule replayed on the oracle (C sweep + the Python restatement
    of the two randomised steps under the same injected streams).  Returns (model, oracle, hyper state) after `iters`."""
    import math
    from hostmirror.binding import FastQMVWVParallelTopicModel
    from oracle.binding import Oracle
    from oracle import dp_samplers as dps
    from mvtopicmodel_amd import synth
    M = len(V)
    c = synth.generate(K, V, D, lam, seed=cseed, chunk_docs=4096)
    training = [(np.arange(c.D, dtype=np.int64), c.doc_off[m], c.tokens[m], V[m]) for m in range(M)]
    model = FastQMVWVParallelTopicModel(K, M, alpha, 0.01)
    model.setNumIterations(iters); model.setBurninPeriod(burnin); model.setOptimizeInterval(interval); model.setRandomSeed(seed)
    model.setNumShards(shards)
    model.addInstances(training)

    o = Oracle(K, V)
    for m in range(M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m])
    hy = Hyper.defaults(K, V, p_a=0.2)
    hy.alpha[:] = alpha; hy.alpha_sum[:] = K * alpha
    inactive = np.zeros(K, dtype=np.uint8)
    push = lambda: o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, inactive)
    push()
    o.init_assignments(seed)
    o.build_counts()
    present = [c.D] * M       # every view lists every entity (some with an empty FeatureSequence): totalDocsPerModality PTM:624
    max_type_count = [int(np.bincount(c.tokens[m], minlength=V[m]).max()) for m in range(M)]
    hist_len = [int(np.diff(c.doc_off[m]).max()) + 1 for m in range(M)]
    # the host mirror seeds its stand-ins for the reference's unseedable streams from randomSeed: samp <- seed+1, random <- seed
    dp = dps.DPState(K, M, hy.alpha, hy.gamma)
    statics, samp, rnd = dps.StaticSamplers(), dps.RandomSamplers(dps.JavaRandom(seed + 1)), dps.JavaRandom(seed)
    state = dict(n_opt=0, ll=None, activations=0, inactive_seen=0)

    def optimize_round():
        sums = o.optimize_p_sums()                                   # PTM:2784-2812
        for m in range(M):
            for i in range(m + 1, M):
                pmean = sums[m, i] / min(present[m], present[i])
                a = 5000 if pmean == 1 else (0.0 if pmean == 0 else -1.0 / math.log(pmean))     # Java: -1.0 / log(0) = -1.0 / -Infinity = 0.0
                hy.p_a[m, i] = hy.p_a[i, m] = min(a, 100.0)
                hy.p_b[m, i] = hy.p_b[i, m] = 1.0
        hists = [o.get_doc_topic_hist(m, hist_len[m], hist_len[m]) for m in range(M)]
        dp.alpha = [list(map(float, a)) for a in hy.alpha]; dp.gamma = list(map(float, hy.gamma))
        dp.inactive = set(np.flatnonzero(inactive).tolist())
        dps.optimize_dp(dp, [h[0] for h in hists], statics, rnd)     # PTM:1184
        dps.optimize_gamma(dp, [h[1] for h in hists], samp)         # PTM:1185
        hy.alpha[:] = np.array(dp.alpha); hy.alpha_sum[:] = dp.alphaSum; hy.gamma[:] = dp.gamma
        inactive[:] = 0; inactive[sorted(dp.inactive)] = 1
        state["inactive_seen"] = max(state["inactive_seen"], int(inactive.sum()))
        for m in range(M):                                           # PTM:2293-2366
            b, bs = o.optimize_beta(m, max_type_count[m])
            hy.beta[m], hy.beta_sum[m] = b, bs

    for it in range(1, iters + 1):
        if it < burnin and M > 1:
            hy.p_a[:] = min(it / 100 + 0.3, 1.1)
        elif it > burnin and interval != 0 and it % interval == 0:
            state["n_opt"] += 1
            optimize_round()
        push()
        r = o.sweep(it, seed)
        if r["stats"]["activated_topic"] >= 0:                       # UPD:263-270 changed alpha / the inactive set
            hy.alpha[:] = o.get_alpha(); inactive[:] = o.get_inactive()
            state["activations"] += 1
        if it % 10 == 0:
            state["ll"] = o.model_log_likelihood()
    model.estimate()
    for m in range(M):
        assert np.array_equal(model.get_view(m)[3], o.get_assignments(m)), f"assignments differ in view {m}"
        a, b = model.get_counts(m)
        assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])
    return model, o, c, hy, inactive, dp, statics, samp, rnd, optimize_round, state, max_type_count, present, hist_len


def test_estimate_with_optimize_steps_matches_oracle_schedule():
    """Past burn-in, every optimizeInterval iterations estimate() runs optimizeP, optimizeDP, optimizeGamma and
    optimizeBeta (PTM:1173-1210) before the sweep; the same schedule replayed on the oracle (C sweep + the
    Python restatement of the two randomised steps, same injected streams) gives the same integers and the
    same hyper-parameters bit for bit."""
    from oracle import dp_samplers as dps
    K, V = 30, [400, 50, 40]
    model, o, c, hy, inactive, dp, statics, samp, rnd, optimize_round, state, max_type_count, present, hist_len = \
        _run_schedule(K, V, 150, [40, 6, 5], 321, 5, 10, 2, 2)
    assert state["n_opt"] == 4
    # one more round of every step, state read back: same numbers as the oracle's next round
    pa, pm = model.optimizeP()
    alpha, asum, ina, tables = model.optimizeDP()
    g, gv, groot = model.optimizeGamma()
    bb, bbs = model.optimizeBeta()
    hists = [o.get_doc_topic_hist(m, hist_len[m], hist_len[m]) for m in range(3)]
    dp.alpha = [list(map(float, a)) for a in hy.alpha]; dp.gamma = list(map(float, hy.gamma))
    dp.inactive = set(np.flatnonzero(inactive).tolist())
    dps.optimize_dp(dp, [h[0] for h in hists], statics, rnd)
    assert np.array_equal(alpha, np.array(dp.alpha)) and np.array_equal(asum, np.array(dp.alphaSum))
    assert set(np.flatnonzero(ina).tolist()) == dp.inactive
    assert tables.tolist() == dp.tablesCnt + [dp.rootTablesCnt]
    dps.optimize_gamma(dp, [h[1] for h in hists], samp)
    assert g.tolist() == dp.gamma and gv.tolist() == dp.gammaView and groot == dp.gammaRoot
    assert np.all(g > 0) and np.all(g != 1.0) and abs(asum.sum() - 3) < 1e-9      # each view's alpha is a Dirichlet mean
    for m in range(3):
        ob, obs = o.optimize_beta(m, max_type_count[m])
        assert bb[m] == ob and bbs[m] == obs
        assert bb[m] != 0.01
    sums = o.optimize_p_sums()
    assert pm[0, 1] == sums[0, 1] / min(present[0], present[1])
    # LL/token recorded at iteration 10 (PTM:1302-1303)
    tok = [int(c.doc_off[m][-1]) for m in range(3)]
    for m in range(3):
        per = model.perplexities(m)
        assert len(per) == 2 and abs(per[1] - state["ll"][m] / tok[m]) < 1e-9 * abs(per[1])
    model.close()


def test_estimate_over_document_shards_is_the_single_handle_chain():
    """estimate() with the model kept as three document shards behind an mvhdp_group (setNumShards): burn-in, the four optimise steps
    every second iteration, the log-likelihood every tenth -- every step routed to its mvhdp_group_* counterpart -- replayed on the
    oracle exactly like the single-handle schedule: the same integers, the same hyper-parameters bit for bit, the same LL/token."""
    K, V = 30, [400, 50, 40]
    model, o, c, hy, inactive, dp, statics, samp, rnd, optimize_round, state, max_type_count, present, hist_len = \
        _run_schedule(K, V, 150, [40, 6, 5], 321, 5, 10, 2, 2, shards=3)
    assert state["n_opt"] == 4
    alpha, asum, ina, tables = model.optimizeDP()
    from oracle import dp_samplers as dps
    hists = [o.get_doc_topic_hist(m, hist_len[m], hist_len[m]) for m in range(3)]
    dp.alpha = [list(map(float, a)) for a in hy.alpha]; dp.gamma = list(map(float, hy.gamma))
    dp.inactive = set(np.flatnonzero(inactive).tolist())
    dps.optimize_dp(dp, [h[0] for h in hists], statics, rnd)
    assert np.array_equal(alpha, np.array(dp.alpha)) and np.array_equal(asum, np.array(dp.alphaSum))
    bb, bbs = model.optimizeBeta()
    for m in range(3):
        ob, obs = o.optimize_beta(m, max_type_count[m])
        assert bb[m] == ob and bbs[m] == obs
    pa, pm = model.optimizeP()
    sums = o.optimize_p_sums()
    assert pm[0, 1] == sums[0, 1] / min(present[0], present[1]) and pm[1, 2] == sums[1, 2] / min(present[1], present[2])
    tok = [int(c.doc_off[m][-1]) for m in range(3)]
    for m in range(3):
        per = model.perplexities(m)
        assert len(per) == 2 and abs(per[1] - state["ll"][m] / tok[m]) < 1e-9 * abs(per[1])
    model.close()


@pytest.mark.parametrize("K,V,D,lam,iters,burnin,interval", [
    (60, [120], 40, [12], 12, 3, 3),                     # one view; far more topics than the corpus can hold: optimizeDP
                                                         #   leaves topics inactive and the new-topic branch brings some back
    (8, [300, 30], 120, [25, 4], 9, 2, 2),               # two views, few topics
    (200, [500, 60, 40, 30, 20], 30, [60, 5, 4, 3, 2], 8, 1, 3),   # five views, entities with a few hundred tokens
])
def test_estimate_schedule_on_more_shapes(K, V, D, lam, iters, burnin, interval):
    model, o, c, hy, inactive, dp, statics, samp, rnd, optimize_round, state, *_ = \
        _run_schedule(K, V, D, lam, 900 + K, 7, iters, burnin, interval)
    assert state["n_opt"] >= 2
    if K == 60:
        assert state["inactive_seen"] > 0                # optimizeDP did find topics without documents
    model.close()


def test_print_state_format(tmp_path):
    """the text state of printState (PTM:3276-3320), plain and gzipped."""
    import gzip
    from hostmirror.binding import FastQMVWVParallelTopicModel, java_double_to_string
    K, V = 7, [30, 9]
    rng = np.random.RandomState(8)
    lens0, lens1 = [3, 2, 4], [1, 2, 1]
    off0 = np.concatenate([[0], np.cumsum(lens0)]).astype(np.int64); off1 = np.concatenate([[0], np.cumsum(lens1)]).astype(np.int64)
    tok0 = rng.randint(0, 30, off0[-1]).astype(np.int32); tok1 = rng.randint(0, 9, off1[-1]).astype(np.int32)
    model = FastQMVWVParallelTopicModel(K, 2, 0.1, 0.01)
    model.setNumIterations(2); model.setRandomSeed(3)
    model.addInstances([(np.arange(3, dtype=np.int64), off0, tok0, 30), (np.arange(3, dtype=np.int64), off1, tok1, 9)])
    model.estimate()
    p_txt, p_gz = tmp_path / "state.txt", tmp_path / "state.txt.gz"
    model.printState(p_txt); model.printState(p_gz)
    text = open(p_txt).read()
    assert gzip.open(p_gz, "rt").read() == text
    lines = text.split("\n")
    assert lines[0] == "#doc source pos typeindex type topic"
    assert lines[1] == "#alpha : modality:0"
    # gamma*alpha = 0.1 for every topic, then the next view's header on the same line (print, not println)
    assert lines[2] == "0.1 " * K + "modality:1"
    assert lines[3] == "0.1 " * K
    assert lines[4] == "#beta[0] : 0.01"
    body = [l.split(" ") for l in lines[5:] if l]
    assert len(body) == int(off0[-1] + off1[-1])
    z0, z1 = model.get_view(0)[3], model.get_view(1)[3]
    # entity-major, view-minor, position order; "NA" source; type printed for the alphabet entry
    exp = []
    for d in range(3):
        for pi in range(lens0[d]):
            i = off0[d] + pi; exp.append([str(d), "NA", str(pi), str(tok0[i]), str(tok0[i]), str(z0[i])])
        for pi in range(lens1[d]):
            i = off1[d] + pi; exp.append([str(d), "NA", str(pi), str(tok1[i]), str(tok1[i]), str(z1[i])])
    assert body == exp
    assert java_double_to_string(1e-4) == "1.0E-4" and java_double_to_string(0.001) == "0.001"
    assert java_double_to_string(1e7) == "1.0E7" and java_double_to_string(9999999.0) == "9999999.0"
    model.close()


def test_print_state_round_trip(tmp_path):
    """write -> read -> the same model without a JVM.  The state a chain wrote (host mirror printState, gzip) is read
    back by mvtopicmodel_amd.state_io (the reference's commented-out initializeFromState, PTM:534-573): identical assignments,
    the header's gamma*alpha, and -- pushed through the C ABI into a fresh handle -- identical counts and the same next sweep."""
    from mvtopicmodel_amd import NativeSampler
    from hostmirror.binding import FastQMVWVParallelTopicModel
    from mvtopicmodel_amd.state_io import read_state
    K, V, D = 12, [80, 20, 15], 40
    c = small_corpus(K, V, D, [18, 3, 4], 77)
    views = []
    for m in range(3):
        off = c.doc_off[m].copy()
        if m > 0:                                                           # printState needs every view present (PTM:3291): give empty views one token
            lens = np.maximum(np.diff(off), 1)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        rng = np.random.RandomState(70 + m)
        tok = rng.randint(0, V[m], off[-1]).astype(np.int32)
        views.append((np.arange(D, dtype=np.int64), off, tok, V[m]))
    model = FastQMVWVParallelTopicModel(K, 3, 0.1, 0.01)
    model.setNumIterations(6); model.setRandomSeed(5)
    model.addInstances(views)
    model.estimate()
    path = tmp_path / "state.gz"
    model.printState(path)
    st = read_state(path, [v[1] for v in views], [v[2] for v in views])
    for m in range(3):
        assert np.array_equal(st["z"][m], model.get_view(m)[3])
    assert st["beta0"] == 0.01 and st["gamma_alpha"].shape == (3, K)
    # a fresh handle from the file alone
    s = NativeSampler(K, V)
    for m in range(3):
        s.set_corpus(m, views[m][1], views[m][2]); s.set_assignments(m, st["z"][m])
    hy = Hyper.defaults(K, V)
    s.set_hyper(hy); s.build_counts()
    for m in range(3):
        nwk, nk = s.get_counts(m)
        ref = np.zeros((V[m], K), dtype=np.int32); np.add.at(ref, (views[m][2], st["z"][m]), 1)
        assert np.array_equal(nwk, ref) and np.array_equal(nk, ref.sum(axis=0))
    # a corpus that does not match is refused, like the reference's reader (PTM:557-559)
    bad = [v[2].copy() for v in views]; bad[0][5] = (bad[0][5] + 1) % V[0]
    with pytest.raises(ValueError):
        read_state(path, [v[1] for v in views], bad)
    s.close(); model.close()


def test_number_format_and_display_top_words():
    """displayTopWords PTM:1852-1890: per topic and view `topic<TAB>alpha<TAB>` then the numWords-1 most frequent types,
    ordered as MALLET's IDSorter orders them (count descending, equal counts by descending type id), alpha through
    NumberFormat with at most five fraction digits."""
    from hostmirror.binding import FastQMVWVParallelTopicModel, number_format5
    for v, want in [(0.1, "0.1"), (1234567.891234, "1,234,567.89123"), (0.000004, "0"), (0.000005, "0.00001"), (2.5e-6, "0"),
                    (12.0, "12"), (1000.0, "1,000"), (-0.5, "-0.5"), (999999.999999, "1,000,000"),
                    (0.123455, "0.12345")]:      # the double below 0.123455: JDK >= 8 rounds the exact binary value (JDK-7131459)
        assert number_format5(v) == want, (v, number_format5(v), want)
    K, V = 6, [40, 9]
    rng = np.random.RandomState(2)
    lens = [rng.randint(5, 30, 25), rng.randint(0, 4, 25)]
    training = []
    for m in range(2):
        off = np.concatenate([[0], np.cumsum(lens[m])]).astype(np.int64)
        training.append((np.arange(25, dtype=np.int64), off, rng.randint(0, V[m], off[-1]).astype(np.int32), V[m]))
    model = FastQMVWVParallelTopicModel(K, 2, 0.1, 0.01)
    model.setNumIterations(2); model.setRandomSeed(4)
    model.addInstances(training)
    model.estimate()
    text = model.displayTopWords(4)
    counts = [model.get_counts(m)[0] for m in range(2)]
    want = ""
    for topic in range(K):
        for m in range(2):
            order = sorted([w for w in range(V[m]) if counts[m][w, topic] > 0], key=lambda w: (-counts[m][w, topic], -w))
            want += f"{topic}\t0.1\t" + "".join(f"{w}; " for w in order[:3])
        want += "\n"
    assert text == want
    lines = model.displayTopWords(3, usingNewLines=True).splitlines()
    assert lines[0] == "0\t0.1"
    model.close()


def test_inferencer_entry_matches_the_oracle_sequence():
    """SURVEY 8f #3 as one call: getInferencer() (PTM:3457) then inferTopicDistributionsOnNewDocs (INF:114-330) = align
    the views by entity name, trees with leaves p_wt (INF:557-586), initial topics drawn from the trees (INF:169-199,
    out-of-vocabulary tokens stay 0), 10 sweeps with nst = 1 / nut = 0 under p_a = 0.2 (INF:216-219), then the topic
    proportions and text of printDocumentTopics(out, 0.03, -1) (INF:326-329, incl. the carried-over counts of a missing
    view, INF:371-386) -- against the same sequence on the oracle."""
    from hostmirror.binding import FastQMVWVParallelTopicModel, java_double_to_string
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.native import Hyper
    from oracle import doc_topics as dto
    from oracle.binding import Oracle, SWEEP_FROZEN as ORC_FROZEN
    K, V = 16, [220, 40]
    c = synth.generate(K, V, 80, [30, 5], seed=311, chunk_docs=4096)
    training = [(np.arange(c.D, dtype=np.int64) + 1000, c.doc_off[m], c.tokens[m], V[m]) for m in range(2)]
    model = FastQMVWVParallelTopicModel(K, 2, 0.1, 0.01)
    model.setNumIterations(6); model.setBurninPeriod(200); model.setOptimizeInterval(50); model.setRandomSeed(5)
    model.addInstances(training)
    model.estimate()
    counts = [model.get_counts(m) for m in range(2)]

    # new documents: view 0 for entities 100..139; view 1 for every third of them plus five names view 0 never saw
    # (appended as entities without view 0, INF:151-157); some out-of-vocabulary types in view 0
    new = synth.generate(K, V, 45, [25, 4], seed=312, chunk_docs=4096)
    tok0 = new.tokens[0][: new.doc_off[0][40]].copy(); tok0[::13] = V[0] + 5
    off0 = new.doc_off[0][:41].copy()
    names0 = np.arange(100, 140, dtype=np.int64)
    pick = [d for d in range(45) if (d % 3 == 0 or d >= 40) and new.doc_off[1][d + 1] > new.doc_off[1][d]]
    names1 = np.asarray([100 + d if d < 40 else 900 + d for d in pick], dtype=np.int64)
    lens1 = [int(new.doc_off[1][d + 1] - new.doc_off[1][d]) for d in pick]
    off1 = np.concatenate([[0], np.cumsum(lens1)]).astype(np.int64)
    tok1 = np.concatenate([new.tokens[1][new.doc_off[1][d]:new.doc_off[1][d + 1]] for d in pick]).astype(np.int32)
    pmean = np.array([[1.0, 0.35], [0.35, 1.0]]); discr = np.array([1.0, 0.8])
    inf = model.getInferencer(discr_weight=discr, p_mean=pmean)
    inf.setRandomSeed(77)
    text = inf.inferTopicDistributionsOnNewDocs([(names0, off0, tok0), (names1, off1, tok1)])

    # the same, by hand, on the oracle: entity order = view-0 instances, then the unmatched view-1 names
    ids = list(names0) + [n for n in names1 if n >= 900]
    D = len(ids)
    pos = {n: i for i, n in enumerate(ids)}
    L1 = np.zeros(D, dtype=np.int64); chunks = [None] * D
    for j, n in enumerate(names1):
        L1[pos[n]] = lens1[j]; chunks[pos[n]] = tok1[off1[j]:off1[j + 1]]
    e_off = [np.concatenate([off0, np.full(D - 40, off0[-1])]).astype(np.int64), np.concatenate([[0], np.cumsum(L1)]).astype(np.int64)]
    e_tok = [tok0, np.concatenate([ch for ch in chunks if ch is not None]).astype(np.int32)]
    assert inf.num_entities() == D
    for m in range(2):
        gi, go, gt, _ = inf.get_view(m)
        assert gi.tolist() == [int(x) for x in ids] and np.array_equal(go, e_off[m]) and np.array_equal(gt, e_tok[m])
    hyi = Hyper.defaults(K, V, p_a=0.2)
    o = Oracle(K, V)
    for m in range(2):
        o.set_corpus(m, e_off[m], e_tok[m]); o.set_counts(m, *counts[m])
    o.set_hyper(hyi.alpha, hyi.alpha_sum, hyi.beta, hyi.beta_sum, hyi.gamma, hyi.p_a, hyi.p_b, None)
    o.build_inference_trees()
    o.init_assignments_from_trees(77)
    for it in range(1, 11):
        ro = o.sweep(it, 77, flags=ORC_FROZEN)
    stats = inf.iteration_stats()
    assert len(stats) == 10 and stats[-1]["oov_skipped"] == ro["stats"]["oov_skipped"] > 0 and stats[-1]["changed"] == 0
    z = []
    for m in range(2):
        zm = inf.get_view(m)[3]
        assert np.array_equal(zm, o.get_assignments(m)), f"inferred topics differ in view {m}"
        z.append(zm)
    assert (z[0][::13] == 0).all()                                   # OOV tokens keep Java's default 0 and are never sampled
    for m in range(2):                                                # the trained model is untouched
        a, b = model.get_counts(m)
        assert np.array_equal(a, counts[m][0]) and np.array_equal(b, counts[m][1])
    w = [1.0 * pmean[0, 0], discr[1] * pmean[0, 1]]
    want = dto.doc_topic_proportions(K, e_off, z, hyi.alpha, hyi.alpha_sum, hyi.gamma, w)
    assert np.array_equal(inf.doc_topics(), want)
    assert text == dto.print_document_topics(want, [str(int(n)) for n in ids], 0.03, -1, java_double_to_string)
    # the carried-over view: entity 1 has no view 1, so it is scored with entity 0's view-1 counts (INF:371-386)
    assert L1[0] > 0 and L1[1] == 0
    solo = dto.doc_topic_proportions(K, [e_off[0][1:3] - e_off[0][1], np.zeros(2, dtype=np.int64)], [z[0][e_off[0][1]:e_off[0][2]], z[1][:0]],
                                     hyi.alpha, hyi.alpha_sum, hyi.gamma, w)
    assert not np.array_equal(solo[0], want[1])
    inf.close(); model.close()


def test_estimate_with_live_updates_keeps_the_counts_consistent():
    from hostmirror.binding import FastQMVWVParallelTopicModel
    from mvtopicmodel_amd import synth
    K, V = 20, [300, 40, 50]
    c = synth.generate(K, V, 120, [30, 4, 6], seed=91, chunk_docs=4096)
    training = [(np.arange(c.D, dtype=np.int64), c.doc_off[m], c.tokens[m], V[m]) for m in range(3)]
    model = FastQMVWVParallelTopicModel(K, 3, 0.1, 0.01)
    model.setNumIterations(12); model.setRandomSeed(3); model.setLiveUpdates(True, 3)
    model.addInstances(training)
    model.estimate()
    for m in range(3):
        _, off, tok, z = model.get_view(m)
        nwk, nk = model.get_counts(m)
        ref = np.zeros_like(nwk); np.add.at(ref, (tok, z), 1)
        assert np.array_equal(ref, nwk) and np.array_equal(ref.sum(axis=0), nk)
    assert model.perplexities(0)[1] < 0
    model.close()
