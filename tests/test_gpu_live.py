"""MVHDP_SWEEP_LIVE: the reference's own update discipline -- the updater applies every delta to the shared count
arrays while the workers are still sampling (UPD:197-218, racy reads by design PTM:84-87).  A live sweep is not
reproducible run to run, so it is tested the way the reference could be: through what must hold whatever the
interleaving (the counts are exactly the counts of z, nothing negative, every token visited once, the branch counters
add up), through the bit-exact deferred sweep run over the same interleaved segments, and through the statistics
(log-likelihood after n sweeps within a whisker of the deferred sweeps')."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper, SWEEP_FROZEN, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_NO_APPLY
from mvtopicmodel_amd._lib import MvhdpError
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu


def _recount(c, z, K):
    out = []
    for m in range(c.M):
        nwk = np.zeros((c.V[m], K), dtype=np.int64)
        np.add.at(nwk, (c.tokens[m], z[m]), 1)
        out.append(nwk)
    return out


def _check_counts_are_counts_of_z(c, s, K):
    z = [s.get_assignments(m) for m in range(c.M)]
    want = _recount(c, z, K)
    for m in range(c.M):
        nwk, nk = s.get_counts(m)
        assert nwk.min() >= 0 and nk.min() >= 0
        assert np.array_equal(nwk.astype(np.int64), want[m]), f"n_wk is not the count of z in view {m}"
        assert np.array_equal(nk.astype(np.int64), want[m].sum(axis=0)), f"n_k is not the column sum in view {m}"


@pytest.mark.parametrize("overlap", ["1", "0"])
@pytest.mark.parametrize("live16", ["0", "1"])
@pytest.mark.parametrize("nseg", [0, 1, 3, 7])
@pytest.mark.parametrize("K,V,D,lam,cseed", [(20, [300, 40, 50], 200, [30, 4, 6], 31), (200, [3000, 300, 300], 300, [127, 7, 15], 32)])
def test_live_sweep_invariants(K, V, D, lam, cseed, nseg, live16, overlap, monkeypatch):
    """live16 = 1: the sweep's atomics of the light n_wk rows land in the 16-bit mirror (two cells per word), which every kernel of the
    sweep gathers from; 0: atomics and gathers on the 32-bit table.  overlap = 1 (the default): two segments in flight -- the next
    segment's trees are rebuilt and its kernels launched when the current one is nearly through; 0: one segment after the other."""
    monkeypatch.setenv("MVHDP_LIVE16", live16)
    monkeypatch.setenv("MVHDP_LIVE_OVERLAP", overlap)
    c = small_corpus(K, V, D, lam, cseed)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    for it in range(4):
        st = s.sweep(it, 77, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(nseg))
        assert st.tokens == c.total_tokens and st.aborted_docs == 0 and st.oov_skipped == 0
        assert st.new_mass_cnt + st.topic_doc_mass_cnt + st.word_ftree_mass_cnt == st.tokens       # WRK:33-35
        assert 0 < st.changed <= st.tokens
        _check_counts_are_counts_of_z(c, s, K)
    s.close()


@pytest.mark.parametrize("force,mode", [("1", "serial"), ("2", "streams"), ("8", "serial"), ("", "")])
@pytest.mark.parametrize("nseg", [2, 5])
def test_deferred_sweep_over_segments_is_bit_exact(nseg, force, mode, monkeypatch):
    """The segmented launch path itself (interleaved queue segments, per-segment classify / overflow chain) under the
    deferred contract: same integers as the oracle, whatever the number of segments, variant or dispatch mode."""
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    if mode:
        monkeypatch.setenv("MVHDP_FORCE_MODE", mode)
    K, V = 300, [2000, 200, 150]
    c = small_corpus(K, V, 157, [200, 9, 12], 33)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    for it in range(3):
        o.sweep(it, 5)
        st = s.sweep(it, 5, flags=SWEEP_LIVE_SEGMENTS(nseg))
        assert st.tokens == c.total_tokens
        assert_same_state(o, s, c.M)
    s.close()


def test_live_no_apply_returns_this_shards_delta_and_restores_the_snapshot():
    K, V = 50, [600, 80]
    c = small_corpus(K, V, 150, [40, 6], 34)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    before = [s.get_counts(m) for m in range(c.M)]
    st = s.sweep(0, 9, flags=SWEEP_LIVE | SWEEP_NO_APPLY)
    assert st.tokens == c.total_tokens
    for m in range(c.M):                                   # the counts are the sweep-start snapshot again
        a, b = s.get_counts(m)
        assert np.array_equal(a, before[m][0]) and np.array_equal(b, before[m][1])
    with pytest.raises(MvhdpError):                        # the deltas are pending: the next sweep must not drop them
        s.sweep(1, 9)
    s.apply_delta(-1, -1)
    _check_counts_are_counts_of_z(c, s, K)                 # snapshot + (after - before) == counts of the new z
    s.sweep(1, 9)
    _check_counts_are_counts_of_z(c, s, K)
    s.close()


@pytest.mark.parametrize("force", ["", "2", "8"])
def test_live16_with_heavy_and_light_rows(force, monkeypatch):
    """A type with more tokens than a 16-bit cell can count is a HEAVY row: its mirror cells all read 65535 and its counts stay in the
    32-bit table (atomics and gathers); the other rows are light and live in the mirror.  Both kinds in one corpus, every register
    variant reading and updating the mirror, several segments (tree rebuild FROM the mirror), then a deferred sweep on top."""
    monkeypatch.setenv("MVHDP_LIVE16", "1")
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    from mvtopicmodel_amd.synth import Corpus
    K, V, D = 24, [40, 7], 2500
    rng = np.random.RandomState(11)
    lens0 = np.full(D, 200, dtype=np.int64); lens1 = rng.randint(0, 5, D).astype(np.int64)
    off = [np.concatenate([[0], np.cumsum(l)]) for l in (lens0, lens1)]
    t0 = rng.randint(1, 40, off[0][-1]).astype(np.int32)
    t0[rng.rand(len(t0)) < 0.5] = 0                                   # type 0: 250 k tokens, far beyond 65534
    c = Corpus(K, V, off, [t0, rng.randint(0, 7, off[1][-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    nwk = s.get_counts(0)[0]
    assert nwk[0].sum() > 65534 and nwk[1:].sum(axis=1).max() < 65534
    for it in range(3):
        st = s.sweep(it, 21, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(3))
        assert st.tokens == c.total_tokens and 0 < st.changed
        _check_counts_are_counts_of_z(c, s, K)
    # the deferred sweep that follows is the oracle's again (the 32-bit table was brought up to date when the live sweep ended)
    for m in range(c.M):
        o.set_assignments(m, s.get_assignments(m))
    o.build_counts()
    o.sweep(3, 21); s.sweep(3, 21)
    assert_same_state(o, s, c.M)
    # document shards: this shard's delta = after - before, the snapshot restored
    before = [s.get_counts(m) for m in range(c.M)]
    s.sweep(4, 21, flags=SWEEP_LIVE | SWEEP_NO_APPLY | SWEEP_LIVE_SEGMENTS(2))
    for m in range(c.M):
        a, b = s.get_counts(m)
        assert np.array_equal(a, before[m][0]) and np.array_equal(b, before[m][1])
    s.apply_delta(-1, -1)
    _check_counts_are_counts_of_z(c, s, K)
    s.close()


def test_live_flag_errors():
    K, V = 10, [50]
    c = small_corpus(K, V, 20, [8], 35)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(0)])
    s.build_trees()
    with pytest.raises(MvhdpError):
        s.sweep(0, 1, flags=SWEEP_LIVE | SWEEP_FROZEN)
    s.close()


def test_live_inactive_topics_are_activated_at_segment_borders():
    """UPD:263-270 in a live sweep: the first delta of a segment that lands on an inactive topic activates it before the
    next segment starts (and the last segment's at the end of the call), so one sweep can give birth to several topics;
    every activation hands alpha[m][K] to the topic (UPD:268) and the samplers move on to the next inactive index."""
    K, V = 30, [400, 50, 60]
    c = small_corpus(K, V, 90, [25, 4, 6], 36)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[22, 25, 27, 28]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 30.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        z[m][np.isin(z[m], [22, 25, 27, 28])] = 2
    s = make_native(c, hy, z)
    s.set_tuning(live16=1)
    born, most = 0, 0
    for it in range(4):
        ina_before = s.get_alpha()[1].copy()
        st = s.sweep(it, 3, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(4))
        a, ina = s.get_alpha()
        assert int(ina_before.sum()) - int(ina.sum()) == st.activations          # one topic per activation left the inactive set
        if st.activations:
            assert st.activated_topic == int(np.flatnonzero(ina_before)[0])      # the first inactive index goes first (WRK:523-526)
            assert ina[st.activated_topic] == 0 and a[st.activated_modality, st.activated_topic] == 30.0     # UPD:268
            newly = np.flatnonzero(ina_before & ~ina.astype(bool))
            assert np.array_equal(newly, np.flatnonzero(ina_before)[:st.activations])   # in index order, no gaps
        born += st.activations; most = max(most, st.activations)
        _check_counts_are_counts_of_z(c, s, K)
    assert born >= 2 and most >= 2
    s.close()


def test_live_and_deferred_sweeps_reach_the_same_likelihood():
    """Statistics, small scale (the curves at C3 size are in profiles/r02_ll_curves.md): after the same number of sweeps
    from the same start both runs improved a lot, the live run is at least as far as the deferred one (every token of a
    deferred sweep sees sweep-start counts, so it mixes more slowly -- the smaller the corpus the more) and not far from it."""
    K, V = 40, [1500, 200, 200]
    c = small_corpus(K, V, 1500, [60, 6, 8], 37)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    out = {}
    for name, flags in (("deferred", 0), ("live", SWEEP_LIVE)):
        s = make_native(c, hy, z0)
        ll0 = s.model_log_likelihood().sum() / c.total_tokens
        for it in range(40):
            s.sweep(it, 123, flags=flags)
        out[name] = (ll0, s.model_log_likelihood().sum() / c.total_tokens)
        s.close()
    (a0, a1), (b0, b1) = out["deferred"], out["live"]
    assert a0 == b0
    assert a1 > a0 + 0.3 and b1 > b0 + 0.3
    assert b1 > a1 - 0.005 * abs(a1)
    assert abs(a1 - b1) < 0.06 * abs(a1)


def test_live_sweep_over_unassigned_tokens_stays_off_the_16_bit_mirror():
    """A first visit of an unassigned token (z = -1, PTM:63) only ADDS to its row: the row's total grows during the sweep, and a row
    classified light when its tree was built could reach 65535 in a mirror cell (read as "see the 32-bit table") or carry into the
    neighbouring cell of the packed word.  So while some token may be unassigned a live sweep stays on the 32-bit table whatever
    mvhdp_tuning.live16 says: one type with more than 65535 tokens, all unassigned, every count still the count of z afterwards."""
    from mvtopicmodel_amd import NativeSampler
    K, V = 8, [40]
    rng = np.random.default_rng(5)
    D, L = 720, 100
    tok = rng.integers(1, V[0], size=D * L).astype(np.int32)
    tok[rng.random(D * L) < 0.95] = 0                                       # type 0: ~68 400 tokens, far beyond a mirror cell
    assert np.count_nonzero(tok == 0) > 65535
    off = (np.arange(D + 1) * L).astype(np.int64)
    s = NativeSampler(K, V, device=0)
    s.set_corpus(0, off, tok)                                               # z = -1 everywhere
    s.set_hyper(Hyper.defaults(K, V))
    s.build_counts()
    s.set_tuning(live16=1)

    class C1:                                                               # (the shape _check_counts_are_counts_of_z wants)
        M, V, tokens = 1, [40], [tok]
    for it in range(3):
        st = s.sweep(it, 9, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
        assert st.tokens == D * L
        _check_counts_are_counts_of_z(C1, s, K)
    # from the second sweep on nothing is unassigned: the mirror is in use again (type 0 as a heavy row), and a host that hands in
    # assignments with a hole switches it off again
    z = s.get_assignments(0)
    z[7] = -1
    s.set_assignments(0, z); s.build_counts()
    st = s.sweep(5, 9, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
    _check_counts_are_counts_of_z(C1, s, K)
    s.close()


def _longest_first(doc_off):
    """the product's work-queue order: entities by decreasing token count over all views, ties in entity order (mvhdp_api.hip)"""
    tot = sum(np.diff(np.asarray(o)) for o in doc_off)
    return np.argsort(-tot, kind="stable").astype(np.int64)


def _one_wave_against_the_oracle(o, s, order, M, sweeps, nseg, rows, live16, seed):
    for it in range(sweeps):
        ro = o.sweep_live_seq(it, seed, order, nseg=nseg, rows=rows, cell16=live16)["stats"]
        st = s.sweep(it, seed, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(nseg))
        assert (st.tokens, st.changed) == (ro["tokens"], ro["changed"]), f"sweep {it}: {st.changed} changed against the oracle's {ro['changed']}"
        assert (st.new_mass_cnt, st.topic_doc_mass_cnt, st.word_ftree_mass_cnt) == (ro["new_mass_cnt"], ro["topic_doc_mass_cnt"], ro["word_ftree_mass_cnt"])
        assert st.word_ftree_mass_cnt > 0
        assert_same_state(o, s, M)


@pytest.mark.parametrize("rows", [1, 0])
@pytest.mark.parametrize("live16", [0, 1])
@pytest.mark.parametrize("nseg", [1, 3])
@pytest.mark.parametrize("K,V,D,lam,cseed,force", [(40, [500, 60, 50], 150, [40, 5, 6], 41, 1), (200, [3000, 300], 120, [160, 9], 42, 4), (600, [900, 90], 60, [200, 30], 43, 16)])
def test_one_wave_live_sweep_equals_the_sequential_oracle(K, V, D, lam, cseed, force, nseg, live16, rows):
    """The live sweep against the oracle (the live path was only ever compared with itself).  With ONE resident wavefront and
    every chunk's atomics waited for (mvhdp_tuning.single_wave) a live sweep is a sequential algorithm with a defined visibility rule --
    a token sees the n_wk deltas of every earlier 64-token chunk, tokensPerTopic / coefficients / roots of the segment start -- which
    oracle/mvhdp_oracle.c::orc_sweep_live_seq restates (UPD:197-218 applied while WRK:425-590 samples).  Both forms of the tree branch:
    rows = 1 the word's live count row (8 cells a lane on the 16-bit mirror, live16 = 1; 4 on the 32-bit table), rows = 0 stored trees of
    the segment start; every integer must agree: assignments, n_wk, n_k, the branch counters.  K = 600: two batches of the mirror's row."""
    c = small_corpus(K, V, D, lam, cseed)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    # one kernel per segment (a class kernel's end is where its block's tokensPerTopic lands): the primary variant holds every list
    s.set_tuning(live16=live16, single_wave=1, live_rows=rows, force_primary=force)
    _one_wave_against_the_oracle(o, s, _longest_first(c.doc_off), c.M, 3, nseg, rows, live16, 13)
    s.close()


@pytest.mark.parametrize("live16", [0, 1])
def test_one_wave_live_sweep_with_a_heavy_row_equals_the_oracle(live16):
    """... where one type holds more than 65534 tokens: on the mirror (live16 = 1) a heavy word's tree branch walks its stored tree of
    the segment start (its cells pass 16 bits), on the 32-bit table it reads its live row like any other."""
    from mvtopicmodel_amd import NativeSampler
    from oracle.binding import Oracle
    K, V = 8, [40]
    rng = np.random.default_rng(6)
    D, L = 700, 100
    tok = rng.integers(1, V[0], size=D * L).astype(np.int32)
    tok[rng.random(D * L) < 0.95] = 0
    assert np.count_nonzero(tok == 0) > 65535
    off = (np.arange(D + 1) * L).astype(np.int64)
    z0 = rng.integers(0, K, size=D * L).astype(np.int32)
    hy = Hyper.defaults(K, V)
    o = Oracle(K, V)
    o.set_corpus(0, off, tok); o.set_assignments(0, z0)
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    o.build_counts()
    s = NativeSampler(K, V, device=0)
    s.set_corpus(0, off, tok); s.set_assignments(0, z0)
    s.set_hyper(hy); s.build_counts()
    s.set_tuning(live16=live16, single_wave=1, force_primary=1)
    _one_wave_against_the_oracle(o, s, _longest_first([off]), 1, 2, 2, 1, live16, 17)
    s.close()


def test_one_wave_live_sweep_with_unassigned_tokens_and_an_inactive_topic_equals_the_oracle():
    """... with tokens that have no topic yet (z = -1: a first visit only adds; such a sweep stays off the mirror) and a truncated HDP
    (inActiveTopicIndex not empty: the new-topic branch WRK:522-526, a topic activated at the end of the segment whose delta reached it
    first, UPD:263-270, its coefficient zero until then)."""
    K, V = 50, [600, 70]
    c = small_corpus(K, V, 140, [50, 7], 44)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[41, 47]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 30.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        z[m][np.isin(z[m], [41, 47])] = 3
        z[m][::11] = -1
        o.set_assignments(m, z[m])
    o.build_counts()
    s = make_native(c, hy, z)
    s.set_tuning(live16=1, single_wave=1, force_primary=1)               # (asked for, and refused by the plan while a token is unassigned)
    order = _longest_first(c.doc_off)
    acts = 0
    for it in range(3):
        ro = o.sweep_live_seq(it, 21, order, nseg=2, rows=1, cell16=0 if it == 0 else 1)["stats"]
        st = s.sweep(it, 21, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
        assert st.changed == ro["changed"] and st.new_mass_cnt == ro["new_mass_cnt"]
        acts += st.activations
        assert_same_state(o, s, c.M)
        a, ina = s.get_alpha()
        assert np.array_equal(a, o.get_alpha()) and np.array_equal(ina, o.get_inactive())
    assert acts >= 1
    s.close()


@pytest.mark.parametrize("case", range(24))
def test_one_wave_live_sweep_fuzz_against_the_sequential_oracle(case):
    """Random shapes of the live-rows form against the sequential oracle: K from a handful to beyond 1024 (one, two -- the stored first-batch
    mass -- and three or more register batches of a row, on the mirror and on the 32-bit table), one to three views, every register variant
    that holds the corpus' longest entity, one to four segments."""
    rng = np.random.default_rng(7000 + case)
    K = int(rng.choice([8, 33, 64, 100, 255, 256, 257, 400, 512, 513, 600, 777, 1000, 1024, 1025, 1100]))
    M = int(rng.integers(1, 4))
    V = [int(rng.integers(30, 900))] + [int(rng.integers(10, 90)) for _ in range(M - 1)]
    lam = [int(rng.integers(20, 150))] + [int(rng.integers(2, 20)) for _ in range(M - 1)]
    D = int(rng.integers(30, 90))
    c = small_corpus(K, V, D, lam, 7100 + case)
    longest = int(max(sum(int(c.doc_off[m][d + 1] - c.doc_off[m][d]) for m in range(c.M)) for d in range(c.D)))
    force = int(rng.choice([r for r in (1, 2, 4, 8, 16) if 64 * r >= min(longest, K)]))
    live16 = int(rng.integers(0, 2))
    nseg = int(rng.choice([1, 2, 4]))
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    s.set_tuning(live16=live16, single_wave=1, live_rows=1, force_primary=force)
    _one_wave_against_the_oracle(o, s, _longest_first(c.doc_off), c.M, 2, nseg, 1, live16, 100 + case)
    s.close()


def test_a_live_sweep_gives_birth_to_topics_chunk_by_chunk():
    """UPD:263-270 takes a topic out of inActiveTopicIndex with the FIRST delta that reaches it and the samplers then draw the next
    inactive index (WRK:523-526): the reference has every inactive topic of C5 active within its first sweep.  In its live-rows form a
    live sweep does the same chunk by chunk (SweepLaunch::births): ONE segment gives birth to many topics, in index order without gaps,
    each topic's alpha[m][K] going to the view of its first delta; the stored-tree form keeps one birth per segment border."""
    K, V = 60, [500, 60]
    c = small_corpus(K, V, 400, [40, 6], 45)
    inactive = np.zeros(K, dtype=np.uint8); inactive[40:] = 1                 # 20 inactive topics
    hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 50.0        # (a new-topic mass that is drawn often)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        z[m][z[m] >= 40] = 7
    s = make_native(c, hy, z)
    born = []
    for it in range(3):
        ina_before = s.get_alpha()[1].copy()
        st = s.sweep(it, 5, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(1))
        a, ina = s.get_alpha()
        born.append(int(st.activations))
        assert int(ina_before.sum()) - int(ina.sum()) == st.activations
        newly = np.flatnonzero(ina_before.astype(bool) & ~ina.astype(bool))
        assert np.array_equal(newly, np.flatnonzero(ina_before)[:st.activations])          # in index order, no gaps
        if st.activations:
            assert st.activated_topic == int(np.flatnonzero(ina_before)[0])
            assert all((a[:, t] == 50.0).sum() == 1 for t in newly)                        # UPD:268: the view of the topic's first delta, that view only
        _check_counts_are_counts_of_z(c, s, K)
    assert born[0] >= 8 and sum(born) == 20, born                            # one segment, many births; all twenty within three sweeps
    s.close()
    s = make_native(c, hy, z)
    s.set_tuning(live_rows=0)                                                 # the stored-tree form: one birth per segment border
    assert s.sweep(0, 5, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(1)).activations == 1
    assert s.sweep(1, 5, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(3)).activations == 3
    s.close()
