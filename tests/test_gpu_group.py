"""mvhdp_group_*: document shards with the exchange step inside the library (the counterpart of the reference's
in-process queue mesh + barrier, PTM:1042-1049, PTM:1232).  One GPU here: a one-member group goes through the real RCCL
collective (one rank); several members on the one device exercise the sharding itself (their deltas are summed on the device
and enter the collective as one rank) -- both must give the oracle's integers.  The driver runs 2 / 4 / 8 GPUs."""
import numpy as np
import pytest

from mvtopicmodel_amd import NativeGroup, synth
from mvtopicmodel_amd._lib import MvhdpError
from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_NO_APPLY
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu


def _shards(c, hy, z, n):
    """n NativeSamplers over contiguous entity ranges balanced by token count (synth.shard_bounds), global entity ids kept."""
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    out = []
    for lo, hi in synth.shard_bounds(tot, n):
        sub = c.slice_docs(lo, hi)
        zs = [z[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(c.M)]
        out.append(make_native(sub, hy, zs, doc_id_base=lo))
    return out


def _assert_group_equals_oracle(o, shards, c):
    for m in range(c.M):
        assert np.array_equal(np.concatenate([s.get_assignments(m) for s in shards]), o.get_assignments(m)), f"z differs in view {m}"
        nwk, nk = o.get_counts(m)
        for s in shards:
            a, b = s.get_counts(m)
            assert np.array_equal(a, nwk) and np.array_equal(b, nk), f"a replica's counts differ in view {m}"


def test_one_member_group_goes_through_rccl_and_equals_the_plain_sweep():
    K, V = 50, [600, 80]
    c = small_corpus(K, V, 120, [40, 6], 91)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    with NativeGroup([s]) as g:
        info = g.info()
        assert (info.local_members, info.local_devices, info.ranks, info.rccl) == (1, 1, 1, 1) and info.rccl_version > 0
        g.build_counts()
        for it in range(4):
            ro = o.sweep(it, 5)
            st = g.sweep(it, 5)[0]
            assert st.tokens == c.total_tokens and st.changed == ro["stats"]["changed"]
            assert_same_state(o, s, 2)
            assert s.trees_current()                                       # rebuilt row range by row range behind the collective
            o.build_trees()
            for m, w in [(0, 3), (1, 79)]:
                assert np.array_equal(o.get_tree(m, w), s.get_tree(m, w))
        assert g.info().last_exchange_ms > 0
        with pytest.raises(MvhdpError):
            g.sweep(9, 5, flags=SWEEP_NO_APPLY)                            # the group sets that itself
    # the same through the one-process-per-GPU entry: an id, one rank
    uid = NativeGroup.unique_id()
    assert len(uid) == 128
    with NativeGroup.from_rank(s, uid, 0, 1) as g:
        assert g.info().rccl == 1 and g.info().ranks == 1
        for it in range(4, 6):
            o.sweep(it, 5); g.sweep(it, 5)
            assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("n,chunks", [(2, 4), (4, 1), (4, 7)])
def test_members_on_one_device_equal_the_single_handle(n, chunks):
    """n document shards (global entity ids, full model replicas) + the in-library exchange == one handle over all entities == the
    oracle, including the topic the sweep activates (the first delta in entity order wins on every replica, UPD:263-270)."""
    K, V = 60, [700, 90, 70]
    c = small_corpus(K, V, 150, [40, 5, 6], 92)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[52, 57]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 25.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(3)]
    for m in range(3):
        z[m][np.isin(z[m], [52, 57])] = 1
        o.set_assignments(m, z[m])
    o.build_counts()
    shards = _shards(c, hy, z, n)
    with NativeGroup(shards) as g:
        info = g.info()
        assert (info.local_members, info.local_devices, info.ranks) == (n, 1, 1)
        g.set_exchange_chunks(chunks)
        g.build_counts()                                                    # local counts summed over the members
        _assert_group_equals_oracle(o, shards, c)
        acts = 0
        for it in range(4):
            ro = o.sweep(it, 11)
            sts = g.sweep(it, 11)
            assert sum(st.tokens for st in sts) == c.total_tokens
            assert all(st.activated_topic == ro["stats"]["activated_topic"] and st.activated_modality == ro["stats"]["activated_modality"] for st in sts)
            acts += ro["stats"]["activated_topic"] >= 0
            _assert_group_equals_oracle(o, shards, c)
            for s in shards:
                a, ina = s.get_alpha()
                assert np.array_equal(a, o.get_alpha()) and np.array_equal(ina, o.get_inactive())
        assert acts >= 1
    for s in shards:
        s.close()


def test_live_sweeps_of_a_group_keep_the_counts_consistent():
    """MVHDP_SWEEP_LIVE across shards (each replica live for its own entities, one sweep stale for the others': AD-LDA): the global
    counts are exactly the counts of the concatenated assignments, on every replica."""
    K, V = 40, [500, 60]
    c = small_corpus(K, V, 160, [50, 6], 93)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(2)]
    shards = _shards(c, hy, z, 3)
    for s in shards:
        s.set_tuning(live16=1)
    with NativeGroup(shards) as g:
        g.build_counts()
        for it in range(3):
            sts = g.sweep(it, 3, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
            assert sum(st.tokens for st in sts) == c.total_tokens
            for m in range(2):
                zc = np.concatenate([s.get_assignments(m) for s in shards])
                ref = np.zeros((V[m], K), dtype=np.int32); np.add.at(ref, (c.tokens[m], zc), 1)
                for s in shards:
                    a, b = s.get_counts(m)
                    assert a.min() >= 0 and np.array_equal(a, ref) and np.array_equal(b, ref.sum(axis=0))
    for s in shards:
        s.close()


def test_segmented_sweep_across_shards_follows_the_oracle():
    """MVHDP_SWEEP_SEGMENT_APPLY on a group: segment s of EVERY member is swept against the same snapshot, the deltas are exchanged
    and applied, then segment s+1 (n exchanges per sweep).  The oracle follows it: segment s = the union of the members' segments
    (each member cuts its own longest-first order), swept with NO_APPLY, applied, activated -- every integer must agree."""
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_APPLY
    K, V = 60, [700, 90, 70]
    c = small_corpus(K, V, 150, [40, 5, 6], 94)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[50, 55, 58]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 25.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(3)]
    for m in range(3):
        z[m][np.isin(z[m], [50, 55, 58])] = 1
        o.set_assignments(m, z[m])
    o.build_counts()
    n, nseg = 3, 4
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    bounds = synth.shard_bounds(tot, n)
    shards = _shards(c, hy, z, n)
    seg_docs = [[] for _ in range(nseg)]
    for lo, hi in bounds:                                                   # each member's own longest-first order, cut into nseg interleaved segments
        order = lo + np.argsort(-tot[lo:hi], kind="stable")
        for sidx in range(nseg):
            seg_docs[sidx].append(order[sidx::nseg])
    with NativeGroup(shards) as g:
        g.build_counts()
        births = 0
        for it in range(3):
            acts = 0
            for sidx in range(nseg):
                r = o.sweep_list(it, 17, np.sort(np.concatenate(seg_docs[sidx])), flags=ORC_NO_APPLY, want_delta=True)
                o.apply_delta(r["delta_nwk"], r["delta_nk"], r["stats"]["activated_topic"], r["stats"]["activated_modality"])
                acts += r["stats"]["activated_topic"] >= 0
            sts = g.sweep(it, 17, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(nseg))
            assert sum(st.tokens for st in sts) == c.total_tokens
            assert all(st.activations == acts for st in sts)
            births += acts
            _assert_group_equals_oracle(o, shards, c)
            for sh in shards:
                a, ina = sh.get_alpha()
                assert np.array_equal(a, o.get_alpha()) and np.array_equal(ina, o.get_inactive())
        assert births >= 2
    for sh in shards:
        sh.close()


def test_group_create_errors():
    K, V = 10, [50]
    c = small_corpus(K, V, 20, [8], 35)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    a = make_native(c, hy, [o.get_assignments(0)])
    c2 = small_corpus(12, V, 20, [8], 35)
    o2 = make_oracle(c2, Hyper.defaults(12, V))
    b = make_native(c2, Hyper.defaults(12, V), [o2.get_assignments(0)])
    with pytest.raises(MvhdpError):
        NativeGroup([a, b])                                                 # different model shapes
    with pytest.raises(MvhdpError):
        NativeGroup([a, a])                                                 # a handle listed twice
    with pytest.raises(MvhdpError):
        NativeGroup([])
    a.close(); b.close()


def _shards_at(c, hy, z, bounds):
    out = []
    for lo, hi in bounds:
        sub = c.slice_docs(lo, hi)
        zs = [z[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(c.M)]
        out.append(make_native(sub, hy, zs, doc_id_base=lo))
    return out


def test_statistics_of_a_sharded_model_equal_the_single_handle():
    """The steps either side of the sweep (PTM:1173-1210, PTM:1296-1320) for a sharded model: three members on one device against ONE
    handle holding every entity, through a schedule of sweeps, statistics and a change of hyper-parameters (what an optimise step
    does).  Integers exactly; the floating-point sums are the single handle's additions in the single handle's order (one process:
    the running sums travel from member to member), so they are compared for equality too -- and the log-likelihood to 1e-12."""
    K, V = 60, [700, 90, 70]
    c = small_corpus(K, V, 180, [40, 5, 6], 96)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(3)]
    one = make_native(c, hy, z)
    shards = _shards(c, hy, z, 3)
    maxlen = [int(np.diff(c.doc_off[m]).max()) + 1 for m in range(3)]
    with NativeGroup(shards) as g:
        g.build_counts()

        def compare():
            ll1, llg = one.model_log_likelihood(), g.model_log_likelihood()
            assert np.allclose(ll1, llg, rtol=1e-12, atol=0) and np.array_equal(ll1, llg), (ll1, llg)
            assert np.array_equal(one.view_overlap_sums(), g.view_overlap_sums())
            for m in range(3):
                h1, d1 = one.get_doc_topic_hist(m, maxlen[m], maxlen[m])
                hg, dg = g.get_doc_topic_hist(m, maxlen[m], maxlen[m])
                assert np.array_equal(h1, hg) and np.array_equal(d1, dg)
                assert np.array_equal(one.get_count_histogram(m, 300), g.get_count_histogram(m, 300))
                for rnd in (0, 3):
                    q1, w1 = one.gamma_doc_statistics(m, 1.3, 99, rnd)
                    qg, wg = g.gamma_doc_statistics(m, 1.3, 99, rnd)
                    assert q1 == qg and abs(w1 - wg) <= 1e-12 * abs(w1)

        compare()
        for it in range(6):
            one.sweep(it, 21)
            g.sweep(it, 21)
            if it in (1, 4):
                compare()
            if it == 2:                                                     # an optimise step: new hyper-parameters on every replica
                hy2 = Hyper.defaults(K, V)
                hy2.alpha[:, :K] *= 1.7; hy2.alpha_sum[:] = hy2.alpha[:, :K].sum(axis=1)
                hy2.beta[:] = [0.02, 0.015, 0.011]; hy2.beta_sum[:] = hy2.beta * np.array(V); hy2.gamma[:] = [1.2, 0.8, 1.1]
                one.set_hyper(hy2); g.set_hyper(hy2)
        for m in range(3):
            assert np.array_equal(np.concatenate([s.get_assignments(m) for s in shards]), one.get_assignments(m))
    one.close()
    for s in shards:
        s.close()


def test_a_failing_member_fails_the_sweep_for_everybody_and_a_recount_recovers():
    """mvhdp_group_sweep's failure protocol: a member whose sweep cannot start (here: its assignments were replaced behind the counts'
    back) makes the call fail -- after every sweep that WAS begun has been finished and every collective entered -- and
    mvhdp_group_build_counts makes the model consistent again; mvhdp_group_abort does the same on request; a group whose member was
    destroyed says so instead of touching freed memory."""
    K, V = 40, [500, 60]
    c = small_corpus(K, V, 160, [50, 6], 97)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(2)]
    shards = _shards(c, hy, z, 3)
    with NativeGroup(shards) as g:
        g.build_counts()
        o.sweep(0, 4); g.sweep(0, 4)
        _assert_group_equals_oracle(o, shards, c)
        shards[1].set_assignments(0, shards[1].get_assignments(0))          # member 1: counts stale -> its sweep is refused
        with pytest.raises(MvhdpError) as ei:
            g.sweep(1, 4)
        assert "member 1" in str(ei.value)
        with pytest.raises(MvhdpError):
            g.sweep(1, 4)                                                   # still refused: every member now needs the recount
        g.build_counts()                                                    # recount from z on every member + sum
        for m in range(2):
            o.set_assignments(m, np.concatenate([s.get_assignments(m) for s in shards]))
        o.build_counts()
        _assert_group_equals_oracle(o, shards, c)
        o.sweep(2, 4); g.sweep(2, 4)
        _assert_group_equals_oracle(o, shards, c)
        g.abort()
        with pytest.raises(MvhdpError) as ei:
            g.sweep(3, 4)
        assert "abort" in str(ei.value)
        g.build_counts()
        o.sweep(4, 4); g.sweep(4, 4)
        _assert_group_equals_oracle(o, shards, c)
        shards[2].close()                                                   # (the header asks for the other order)
        with pytest.raises(MvhdpError) as ei:
            g.sweep(5, 4)
        assert ei.value.code == -2
    for s in shards[:2]:
        s.close()


def test_segmented_group_sweep_with_a_member_shorter_than_the_segment_count():
    """Every rank must walk through the same number of exchanges: the segment count comes from the flags, a member with fewer entities
    than segments has empty segments (it used to clip the count -- and would have issued fewer collectives than its peers)."""
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_APPLY
    K, V = 30, [400, 50]
    c = small_corpus(K, V, 90, [30, 5], 98)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(2)]
    bounds = [(0, 60), (60, 62), (62, 90)]                                  # the middle member holds two entities
    nseg = 5
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    shards = _shards_at(c, hy, z, bounds)
    seg_docs = [[] for _ in range(nseg)]
    for lo, hi in bounds:
        order = lo + np.argsort(-tot[lo:hi], kind="stable")
        for sidx in range(nseg):
            seg_docs[sidx].append(order[sidx::nseg])
    with NativeGroup(shards) as g:
        g.build_counts()
        for it in range(2):
            for sidx in range(nseg):
                r = o.sweep_list(it, 17, np.sort(np.concatenate(seg_docs[sidx])), flags=ORC_NO_APPLY, want_delta=True)
                o.apply_delta(r["delta_nwk"], r["delta_nk"], -1, -1)
            sts = g.sweep(it, 17, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(nseg))
            assert sum(st.tokens for st in sts) == c.total_tokens
            _assert_group_equals_oracle(o, shards, c)
    for sh in shards:
        sh.close()


def test_async_exchange_keeps_the_global_counts_after_a_drain():
    """MVHDP_SWEEP_ASYNC_EXCHANGE: the all-reduce of a live sweep's deltas runs beside the next sweep.  Between sweeps the replicas
    differ (each lacks the others' last sweep); after mvhdp_group_drain -- or by themselves, the group's statistics -- every replica
    holds exactly the counts of the concatenated assignments.  A sweep without the flag drains first; a deferred sweep refuses it."""
    from mvtopicmodel_amd.native import SWEEP_ASYNC_EXCHANGE
    K, V = 40, [500, 60]
    c = small_corpus(K, V, 160, [50, 6], 99)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(2)]
    shards = _shards(c, hy, z, 3)

    def check_global():
        for m in range(2):
            zc = np.concatenate([s.get_assignments(m) for s in shards])
            ref = np.zeros((V[m], K), dtype=np.int32); np.add.at(ref, (c.tokens[m], zc), 1)
            for s in shards:
                a, b = s.get_counts(m)
                assert a.min() >= 0 and np.array_equal(a, ref) and np.array_equal(b, ref.sum(axis=0))

    with NativeGroup(shards) as g:
        g.build_counts()
        fl = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2) | SWEEP_ASYNC_EXCHANGE
        for it in range(4):
            sts = g.sweep(it, 3, flags=fl)
            assert sum(st.tokens for st in sts) == c.total_tokens
        # in flight: the replicas differ from each other now (each has its own last sweep, not the others')
        a0, a1 = shards[0].get_counts(0)[0], shards[1].get_counts(0)[0]
        assert not np.array_equal(a0, a1)
        g.drain()
        check_global()
        g.sweep(4, 3, flags=fl); g.sweep(5, 3, flags=fl)
        ll = g.model_log_likelihood()                                       # drains by itself
        check_global()
        assert np.all(np.isfinite(ll))
        g.sweep(6, 3, flags=fl)
        g.sweep(7, 3, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))            # a synchronous live sweep: drains first, exchanges at once
        check_global()
        g.sweep(8, 3)                                                       # and a deferred one
        check_global()
        with pytest.raises(MvhdpError):
            g.sweep(9, 3, flags=SWEEP_ASYNC_EXCHANGE)                       # goes with LIVE only
    for s in shards:
        s.close()
