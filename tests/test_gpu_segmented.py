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
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    best = (KEY_NONE, -1, -1)
    stats = dict(tokens=0, changed=0, new_mass_cnt=0, topic_doc_mass_cnt=0, word_ftree_mass_cnt=0)
    for docs in segment_lists(c, nseg):
        r = o.sweep_list(it, seed, docs, flags=ORC_NO_APPLY, want_delta=True)
        o.apply_delta(r["delta_nwk"], r["delta_nk"], -1, -1)
        st = r["stats"]
        for k in stats:
            stats[k] += st[k]
        if st["activated_topic"] >= 0 and st["activation_key"] < best[0]:
            best = (st["activation_key"], st["activated_topic"], st["activated_modality"])
    if best[1] >= 0:                                         # UPD:263-270, at the end of the sweep
        z = np.zeros((sum(o.V), o.K), dtype=np.int32)
        o.apply_delta(z, np.zeros((o.M, o.K), dtype=np.int32), best[1], best[2])
    return stats, best


@pytest.mark.parametrize("force,mode", [("", ""), ("1", "optimistic"), ("2", "classified"), ("8", "optimistic")])
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
        assert (st.activated_topic, st.activated_modality) == (best[1], best[2])
        assert st.new_mass_cnt == so["new_mass_cnt"]
        acts += st.activated_topic >= 0
        assert_same_state(o, s, c.M)
        assert np.array_equal(s.get_alpha()[0], o.get_alpha()) and np.array_equal(s.get_alpha()[1], o.get_inactive())
    assert acts >= 1
    for bad in (SWEEP_LIVE, SWEEP_NO_APPLY):
        with pytest.raises(MvhdpError):
            s.sweep(9, 3, flags=SWEEP_SEGMENT_APPLY | bad)
    s.close()
