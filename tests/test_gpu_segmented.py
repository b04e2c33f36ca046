"""MVHDP_SWEEP_SEGMENT_APPLY: a deferred sweep cut into interleaved segments with the updater catching up in between.
Deterministic, so it is held to the parity bar: the oracle follows it segment by segment (orc_sweep_list over the same
entity lists, trees rebuilt from the current counts before each, deltas applied after each, the topic activation at the
end of the sweep) and every integer must agree."""
import numpy as np
import pytest

from mvtopicmodel_amd._lib import MvhdpError
from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_NO_APPLY, SWEEP_SEGMENT_APPLY
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu
KEY_NONE = (1 << 63) - 1


def segment_lists(c, nseg):
    """The library's segments: positions s, s+n, ... of the entities by decreasing token count, ties by entity index."""
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    order = np.argsort(-tot, kind="stable")
    nseg = max(1, min(nseg, c.D))
    return [order[s::nseg] for s in range(nseg)]


def oracle_segmented_sweep(o, c, it, seed, nseg):
    """The deferred sweep cut into nseg interleaved segments with the updater catching up in between: counts += delta, and
    -- UPD:263-270 -- the segment's first delta on an inactive topic activates it before the next segment starts (the
    samplers then draw the next inactive index, WRK:523-526).  Returns the summed statistics and (key, topic, view,
    activations) of the sweep's first activation."""
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    first = (KEY_NONE, -1, -1)
    n_act = 0
    stats = dict(tokens=0, changed=0, new_mass_cnt=0, topic_doc_mass_cnt=0, word_ftree_mass_cnt=0)
    for docs in segment_lists(c, nseg):
        r = o.sweep_list(it, seed, docs, flags=ORC_NO_APPLY, want_delta=True)
        st = r["stats"]
        o.apply_delta(r["delta_nwk"], r["delta_nk"], st["activated_topic"], st["activated_modality"])
        for k in stats:
            stats[k] += st[k]
        if st["activated_topic"] >= 0:
            if n_act == 0:
                first = (st["activation_key"], st["activated_topic"], st["activated_modality"])
            n_act += 1
    return stats, first + (n_act,)


@pytest.mark.parametrize("force,mode", [("", ""), ("1", "serial"), ("2", "streams"), ("8", "serial")])
@pytest.mark.parametrize("nseg", [0, 2, 5])
def test_segmented_sweep_bit_exact(nseg, force, mode, monkeypatch):
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
        so, _ = oracle_segmented_sweep(o, c, it, 5, nseg or 4)
        st = s.sweep(it, 5, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(nseg))
        assert (st.tokens, st.changed, st.topic_doc_mass_cnt, st.word_ftree_mass_cnt) == \
               (so["tokens"], so["changed"], so["topic_doc_mass_cnt"], so["word_ftree_mass_cnt"])
        assert_same_state(o, s, c.M)
    s.close()


def test_segmented_sweep_with_inactive_topics_and_errors():
    K, V = 30, [400, 50, 60]
    c = small_corpus(K, V, 90, [25, 4, 6], 36)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[25, 28]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 30.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        z[m][np.isin(z[m], [25, 28])] = 2
        o.set_assignments(m, z[m])
    o.build_counts()
    s = make_native(c, hy, z)
    acts = 0
    for it in range(4):
        so, best = oracle_segmented_sweep(o, c, it, 3, 3)
        st = s.sweep(it, 3, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(3))
        assert (st.activated_topic, st.activated_modality, st.activations) == (best[1], best[2], best[3])
        assert st.new_mass_cnt == so["new_mass_cnt"]
        acts += st.activations
        assert_same_state(o, s, c.M)
        assert np.array_equal(s.get_alpha()[0], o.get_alpha()) and np.array_equal(s.get_alpha()[1], o.get_inactive())
    assert acts >= 2                                       # both inactive topics were born
    for bad in (SWEEP_LIVE, SWEEP_NO_APPLY):
        with pytest.raises(MvhdpError):
            s.sweep(9, 3, flags=SWEEP_SEGMENT_APPLY | bad)
    s.close()


def test_segmented_sweep_activates_a_topic_at_every_segment_border():
    """Many inactive topics and a large new-topic weight: every segment draws the new-topic branch often enough to land on
    the current first inactive index, so one sweep gives birth to several topics, one per segment, as the reference's
    updater does while the samplers run (UPD:263-270, WRK:523-526) -- and exactly as the oracle's segmented schedule."""
    K, V = 40, [500, 60]
    c = small_corpus(K, V, 160, [30, 5], 44)
    inactive = np.zeros(K, dtype=np.uint8); inactive[20:] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 50.0
    o = make_oracle(c, hy)
    z = [o.get_assignments(m) % 20 for m in range(c.M)]
    for m in range(c.M):
        o.set_assignments(m, z[m])
    o.build_counts()
    s = make_native(c, hy, z)
    born = 0
    for it in range(3):
        so, best = oracle_segmented_sweep(o, c, it, 11, 6)
        st = s.sweep(it, 11, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(6))
        assert (st.activated_topic, st.activated_modality, st.activations) == (best[1], best[2], best[3])
        assert st.new_mass_cnt == so["new_mass_cnt"] > 0
        born += st.activations
        assert_same_state(o, s, c.M)
        assert np.array_equal(s.get_alpha()[0], o.get_alpha()) and np.array_equal(s.get_alpha()[1], o.get_inactive())
    assert born >= 6 and st.activations >= 2
    s.close()


def oracle_overlapped_sweep(o, c, it, seed, nseg):
    """MVHDP_SWEEP_SEGMENT_OVERLAP: the updater runs beside the samplers -- the deltas of segment s are applied while segment s+1
    is sampled, so segment s samples against the counts after segment s-2 (segments 0 and 1: the sweep-start counts); the F+trees are
    those of the sweep start for every segment (PTM:1209: built in buildFTrees, not per delta)."""
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY, SWEEP_REUSE_TREES as ORC_REUSE_TREES
    stats = dict(tokens=0, changed=0, new_mass_cnt=0, topic_doc_mass_cnt=0, word_ftree_mass_cnt=0)
    pending = []
    o.build_trees()
    for sidx, docs in enumerate(segment_lists(c, nseg)):
        if sidx >= 2:
            r0 = pending.pop(0)
            o.apply_delta(r0["delta_nwk"], r0["delta_nk"], -1, -1)
        r = o.sweep_list(it, seed, docs, flags=ORC_NO_APPLY | ORC_REUSE_TREES, want_delta=True)
        pending.append(r)
        for k in stats:
            stats[k] += r["stats"][k]
    for r0 in pending:
        o.apply_delta(r0["delta_nwk"], r0["delta_nk"], -1, -1)
    return stats


@pytest.mark.parametrize("force,mode", [("", ""), ("1", "serial"), ("2", "streams"), ("8", "serial")])
@pytest.mark.parametrize("nseg", [2, 3, 5, 8])
def test_overlapped_segmented_sweep_bit_exact(nseg, force, mode, monkeypatch):
    """Two segments in flight, the updater's kernel beside the samplers: every integer is the oracle's lag-two schedule's, whatever
    the segment count (even: the model ends in copy 0; odd: in copy 1 and is carried over), variant or dispatch mode -- and the
    16-bit mirror, which follows by packed deltas, is the counts' (a deferred sweep afterwards gathers from it)."""
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_OVERLAP
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    if mode:
        monkeypatch.setenv("MVHDP_FORCE_MODE", mode)
    K, V = 300, [2000, 200, 150]
    c = small_corpus(K, V, 157, [200, 9, 12], 33)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    for it in range(4):
        so = oracle_overlapped_sweep(o, c, it, 5, nseg)
        st = s.sweep(it, 5, flags=SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(nseg))
        assert (st.tokens, st.changed, st.topic_doc_mass_cnt, st.word_ftree_mass_cnt) == \
               (so["tokens"], so["changed"], so["topic_doc_mass_cnt"], so["word_ftree_mass_cnt"])
        assert_same_state(o, s, c.M)
    # a plain deferred sweep and a plain segmented one afterwards: the handle is in an ordinary state
    o.sweep(9, 5); s.sweep(9, 5)
    assert_same_state(o, s, c.M)
    oracle_segmented_sweep(o, c, 10, 5, 3); s.sweep(10, 5, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(3))
    assert_same_state(o, s, c.M)
    s.close()


def test_overlapped_segmented_sweep_small_shapes_oov_and_batches():
    """One view, types outside the alphabet, unassigned tokens at the start (the mirror stays out while row totals can grow), more
    segments than a kernel has waves' worth of entities, and the same sweeps as one mvhdp_sweep_many batch on a twin handle."""
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_OVERLAP
    K, V = 24, [120]
    c = small_corpus(K, V, 70, [18], 61)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z = [o.get_assignments(0).copy()]
    z[0][::7] = -1
    o.set_assignments(0, z[0]); o.build_counts()
    a, b = make_native(c, hy, z), make_native(c, hy, z)
    fl = SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(6)
    for it in range(3):
        oracle_overlapped_sweep(o, c, it, 8, 6)
        a.sweep(it, 8, flags=fl)
        assert_same_state(o, a, 1)
    b.sweep_many(0, 3, 8, flags=fl)
    assert_same_state(o, b, 1)
    with pytest.raises(MvhdpError):
        a.sweep(5, 8, flags=SWEEP_SEGMENT_OVERLAP)                          # goes with SEGMENT_APPLY only
    a.close(); b.close()


def test_overlapped_segmented_sweep_refuses_inactive_topics():
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_OVERLAP
    K, V = 20, [100]
    c = small_corpus(K, V, 40, [10], 62)
    inactive = np.zeros(K, dtype=np.uint8); inactive[18] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    o = make_oracle(c, hy)
    z = [o.get_assignments(0) % 18]
    s = make_native(c, hy, z)
    with pytest.raises(MvhdpError) as ei:
        s.sweep(0, 1, flags=SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(4))
    assert ei.value.code == -6
    s.close()
