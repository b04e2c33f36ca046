"""Parity proper: the HIP sweep (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json north_star): integer topic-word /
doc-topic counts and assignments bit-exact after a sweep under the identical RNG
stream; per-token conditional probabilities within 1e-6."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import (Hyper, SWEEP_EXACT_CHAIN, SWEEP_GENERIC_KERNEL, SWEEP_NO_APPLY, SWEEP_REUSE_TREES)
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-6     # north_star: per-token conditional probabilities within 1e-6

CASES = [
    # K, V, D, lam, corpus seed     (K deliberately not powers of two, plus one that is)
    (5, [40], 64, [12], 11),
    (20, [300, 40, 50], 64, [30, 4, 6], 12),
    (100, [2000], 96, [127], 13),
    (64, [500, 60], 64, [40, 5], 14),
    (200, [3000, 300, 300], 80, [127, 7, 15], 15),
    (400, [5000, 500, 500], 48, [127, 7, 15], 16),
]


KERNELS = [0, SWEEP_GENERIC_KERNEL]     # register-resident kernel (default) and the generic LDS kernel


@pytest.mark.parametrize("kflag", KERNELS)
@pytest.mark.parametrize("K,V,D,lam,cseed", CASES)
def test_one_sweep_bit_exact(K, V, D, lam, cseed, kflag):
    c = small_corpus(K, V, D, lam, cseed)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    s = make_native(c, hy, z0)
    assert_same_state(o, s, c.M)                     # build_counts parity
    ro = o.sweep(0, 0xC0FFEE, want_dbg=True)
    rs = s.sweep(0, 0xC0FFEE, flags=kflag, want_dbg=True)
    for f in ("tokens", "changed", "new_mass_cnt", "topic_doc_mass_cnt", "word_ftree_mass_cnt", "oov_skipped", "aborted_docs"):
        assert ro["stats"][f] == getattr(rs, f), f
    assert rs.tokens == c.total_tokens
    assert_same_state(o, s, c.M)
    if c.M > 1:
        assert np.array_equal(o.draw_p_philox(0xC0FFEE, 0), s.get_view_weights())
    for m in range(c.M):
        # masses: newTopicMass / tree root bit-exact; topicDocWordMass within the scan's reordering error
        a, b = ro["dbg"][m], rs.dbg[m]
        assert np.array_equal(a[:, 0], b[:, 0])
        assert np.array_equal(a[:, 2], b[:, 2])
        assert np.allclose(a[:, 1], b[:, 1], rtol=1e-12, atol=0)
        assert np.allclose(a[:, 3], b[:, 3], rtol=1e-12, atol=0)
    s.close()


@pytest.mark.parametrize("kflag", KERNELS)
@pytest.mark.parametrize("K,V,D,lam,cseed", CASES[1:5])
def test_three_sweeps_bit_exact_and_exact_chain_mode(K, V, D, lam, cseed, kflag):
    """Several sweeps (state carried on the device), and the forced sequential-sum
    mode (the certified scan's fallback path) must give the very same integers."""
    c = small_corpus(K, V, D, lam, cseed)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    s = make_native(c, hy, z0)
    s2 = make_native(c, hy, z0)
    for it in range(3):
        ro = o.sweep(it, 42, want_dbg=True)
        s.sweep(it, 42, flags=kflag)
        r2 = s2.sweep(it, 42, flags=SWEEP_EXACT_CHAIN | kflag, want_dbg=True)
        assert_same_state(o, s, c.M)
        assert_same_state(o, s2, c.M)
        for m in range(c.M):
            # in exact-chain mode every mass is bit-identical to the oracle's
            assert np.array_equal(ro["dbg"][m], r2.dbg[m])
    s.close(); s2.close()


def test_trees_match_oracle_bitwise():
    K, V = 200, [700, 90]
    c = small_corpus(K, V, 50, [60, 6], 21)
    hy = Hyper.defaults(K, V)
    hy.alpha[:] = np.linspace(0.01, 0.3, K + 1)[None, :]       # asymmetric alpha
    hy.alpha_sum[:] = hy.alpha[:, :K].sum(axis=1)
    hy.gamma[:] = [1.0, 0.7]
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    o.build_trees(); s.build_trees()
    for m, w in [(0, 0), (0, 1), (0, 699), (1, 0), (1, 89), (0, 345)]:
        assert np.array_equal(o.get_tree(m, w), s.get_tree(m, w)), (m, w)
    s.close()


@pytest.mark.parametrize("kflag", KERNELS)
def test_token_conditionals_within_1e6(kflag):
    K, V = 100, [1500, 200, 200]
    c = small_corpus(K, V, 40, [60, 6, 9], 31)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(3)])
    trace = []
    for d in range(0, 40, 3):
        for m in range(3):
            L = int(c.doc_off[m][d + 1] - c.doc_off[m][d])
            for pos in {0, L // 2, L - 1}:
                if 0 <= pos < L:
                    trace.append((d, m, pos))
    ro = o.sweep(0, 5, trace=trace)
    rs = s.sweep(0, 5, flags=kflag, trace=trace)
    assert ro["trace"].shape == rs.trace.shape
    assert np.allclose(ro["trace"].sum(axis=1), 1.0, atol=1e-9)
    assert np.max(np.abs(ro["trace"] - rs.trace)) < PROB_TOL
    assert_same_state(o, s, 3)
    s.close()


def test_p_override_from_mallet_stream_and_reuse_trees():
    """Host-drawn view weights (the reference worker's own MALLET Randoms, WRK:327-337) passed as an override."""
    K, V = 50, [800, 100, 100]
    c = small_corpus(K, V, 60, [40, 5, 8], 41)
    hy = Hyper.defaults(K, V, p_a=1.1)                          # exercises the a>=1,b==1 quirk branch too
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(3)])
    p = o.draw_p_mallet(seed=1)
    o.sweep(0, 9, p=p)
    s.sweep(0, 9, p=p)
    assert_same_state(o, s, 3)
    assert np.array_equal(s.get_view_weights(), p)
    # stale trees on request: the counts moved, the trees did not
    s.build_trees(); o.build_trees()
    from oracle.binding import SWEEP_REUSE_TREES as ORT
    o.sweep(1, 9, p=p, flags=ORT)
    s.sweep(1, 9, p=p, flags=SWEEP_REUSE_TREES)
    assert_same_state(o, s, 3)
    s.close()


@pytest.mark.parametrize("kflag", KERNELS)
def test_ragged_empty_unassigned_oov(kflag):
    """Edge cases the reference handles: entities missing a view (null), empty entities,
    UNASSIGNED topics (-1, PTM:63), out-of-vocabulary types (WRK:427-428)."""
    K, V = 30, [100, 20]
    rng = np.random.RandomState(0)
    lens0 = np.array([0, 5, 1, 0, 70, 3, 0, 9, 2, 65], dtype=np.int64)
    lens1 = np.array([0, 0, 4, 2, 0, 1, 0, 3, 0, 10], dtype=np.int64)
    off0 = np.concatenate([[0], np.cumsum(lens0)]); off1 = np.concatenate([[0], np.cumsum(lens1)])
    t0 = rng.randint(0, 100, off0[-1]).astype(np.int32); t1 = rng.randint(0, 20, off1[-1]).astype(np.int32)
    t0[[3, 20]] = 100                                           # OOV: type == V (inference-only path)
    t1[2] = 25
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, [off0, off1], [t0, t1])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    z0[0][[0, 7, 30]] = -1; z0[1][[1, 5]] = -1                  # UNASSIGNED
    for m in range(2):
        o.set_assignments(m, z0[m])
    # the OOV tokens must not be counted by build_counts either: give them UNASSIGNED like a fresh inference doc
    z0[0][[3, 20]] = -1; z0[1][2] = -1
    for m in range(2):
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    ro = o.sweep(0, 77); rs = s.sweep(0, 77, flags=kflag)
    assert ro["stats"]["oov_skipped"] == rs.oov_skipped == 3
    assert rs.tokens == c.total_tokens - 3
    assert_same_state(o, s, 2)
    assert (s.get_assignments(0)[[3, 20]] == -1).all()
    for it in range(1, 3):
        o.sweep(it, 77); s.sweep(it, 77, flags=kflag)
        assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("kflag", KERNELS)
def test_inactive_topic_activation(kflag):
    """Truncated-HDP branch: a non-empty inActiveTopicIndex gives newTopicMass>0 (WRK:515-526);
    the first delta that lands on the inactive topic activates it (UPD:263-270)."""
    K, V = 40, [300, 50]
    c = small_corpus(K, V, 80, [30, 5], 51)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[33, 36, 39]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 25.0                                         # make the new-topic branch likely
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    for m in range(2):                                            # inactive topics hold no tokens
        z0[m][np.isin(z0[m], [33, 36, 39])] = 1
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    ro = o.sweep(0, 3, want_dbg=True); rs = s.sweep(0, 3, flags=kflag, want_dbg=True)
    assert ro["stats"]["new_mass_cnt"] == rs.new_mass_cnt > 0
    assert (ro["stats"]["activated_topic"], ro["stats"]["activated_modality"]) == (rs.activated_topic, rs.activated_modality)
    assert rs.activated_topic == 33
    assert_same_state(o, s, 2)
    a_s, ina_s = s.get_alpha()
    assert np.array_equal(o.get_alpha(), a_s) and np.array_equal(o.get_inactive(), ina_s)
    assert ina_s[33] == 0 and ina_s[36] == 1
    assert a_s[rs.activated_modality, 33] == 25.0
    ro = o.sweep(1, 3); rs = s.sweep(1, 3, flags=kflag)
    assert rs.activated_topic == 36
    assert_same_state(o, s, 2)
    s.close()


def test_doc_topic_hist_matches():
    K, V = 60, [500, 60]
    c = small_corpus(K, V, 70, [50, 6], 61)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    o.sweep(0, 1); s.sweep(0, 1)
    for m in range(2):
        ho, lo = o.get_doc_topic_hist(m, 128, 128)
        hs, ls = s.get_doc_topic_hist(m, 128, 128)
        assert np.array_equal(ho, hs) and np.array_equal(lo, ls)
        # a histogram shorter than the largest per-entity count: an entity holding a topic more often than the last
        # bucket is in no bucket -- in particular not in bucket 0 (non-holders, PTM:647-649)
        ho, lo = o.get_doc_topic_hist(m, 3, 5)
        hs, ls = s.get_doc_topic_hist(m, 3, 5)
        assert np.array_equal(ho, hs) and np.array_equal(lo, ls)
    assert o.get_doc_topic_hist(0, 128, 0)[0][:, 3:].sum() > 0      # the short histogram really cut something off
    s.close()


def test_no_apply_then_apply_equals_apply():
    K, V = 50, [400, 60]
    c = small_corpus(K, V, 64, [40, 6], 71)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    s = make_native(c, hy, z0)
    before = [s.get_counts(m) for m in range(2)]
    st = s.sweep(0, 8, flags=SWEEP_NO_APPLY)
    for m in range(2):
        assert np.array_equal(before[m][0], s.get_counts(m)[0])
    s.apply_delta(st.activated_topic, st.activated_modality)
    o.sweep(0, 8)
    assert_same_state(o, s, 2)
    s.close()


def test_long_entities_take_the_generic_kernel_and_work_queue_order():
    """Entities with > 256 tokens and K > 256 exceed the register-resident kernel's slot budget;
    very uneven lengths also switch the work queue to longest-first order."""
    K, V = 500, [3000, 200]
    rng = np.random.RandomState(5)
    lens0 = np.array([900, 3, 40, 700, 1, 0, 350, 12, 5, 1200, 64, 65, 128, 129, 2], dtype=np.int64)
    lens1 = np.array([10, 0, 4, 30, 0, 2, 6, 0, 1, 25, 3, 3, 0, 7, 1], dtype=np.int64)
    off0 = np.concatenate([[0], np.cumsum(lens0)]); off1 = np.concatenate([[0], np.cumsum(lens1)])
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, [off0, off1], [rng.randint(0, 3000, off0[-1]).astype(np.int32), rng.randint(0, 200, off1[-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    for it in range(3):
        ro = o.sweep(it, 17); rs = s.sweep(it, 17)
        assert rs.tokens == c.total_tokens
        assert_same_state(o, s, 2)
    s.close()


def test_overflow_entities_are_rerun_by_the_generic_kernel():
    """After the first sweep the register-resident kernel is sized for the bulk of the entities
    (<= 0.5% overflow); the rare entity with a longer topic list goes through the overflow list
    to the generic kernel inside the same mvhdp_sweep call.  Results must not change."""
    K, V = 400, [3000]
    rng = np.random.RandomState(9)
    lens = np.full(601, 12, dtype=np.int64)
    lens[[100, 377]] = [330, 200]                     # ~225 and ~160 distinct topics: need 4 and 3 slot rounds
    off = np.concatenate([[0], np.cumsum(lens)])
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, [off], [rng.randint(0, 3000, off[-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(0)])
    for it in range(4):
        ro = o.sweep(it, 23); rs = s.sweep(it, 23)
        assert rs.tokens == c.total_tokens == ro["stats"]["tokens"]
        assert rs.changed == ro["stats"]["changed"]
        assert_same_state(o, s, 1)
    s.close()


@pytest.mark.parametrize("mode", ["serial", "streams", ""])
@pytest.mark.parametrize("force", ["", "1", "2", "4", "8", "16"])
def test_register_variants_overflow_chain_and_classified_side_streams(force, mode, monkeypatch):
    """Topic lists of ~50 ... ~1500 distinct topics in one corpus: whatever primary variant (64*r slots,
    r = 1..16) the sweep starts with, the entities that do not fit reach a wider variant and, beyond
    1024 slots, the generic LDS kernel -- either one pass after another through overflow lists
    ("serial") or measured up front and run side by side on their own streams ("streams");
    all inside one mvhdp_sweep call, with identical results."""
    if mode:
        monkeypatch.setenv("MVHDP_FORCE_MODE", mode)
    K, V = 2048, [5000, 300]
    rng = np.random.RandomState(21)
    lens0 = np.array([60, 110, 150, 260, 300, 520, 700, 1100, 1500, 2600, 4000, 9, 0, 33, 64, 128, 256, 512, 1024] + [20] * 40, dtype=np.int64)
    lens1 = rng.randint(0, 12, len(lens0)).astype(np.int64)
    off0 = np.concatenate([[0], np.cumsum(lens0)]); off1 = np.concatenate([[0], np.cumsum(lens1)])
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, [off0, off1], [rng.randint(0, 5000, off0[-1]).astype(np.int32), rng.randint(0, 300, off1[-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    d0 = np.unique(z0[0][off0[10]:off0[11]]).size
    assert d0 > 1024                                    # the 4000-token entity really needs the generic pass
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    s = make_native(c, hy, z0)
    for it in range(4):
        ro = o.sweep(it, 31)
        rs = s.sweep(it, 31, flags=SWEEP_EXACT_CHAIN if it == 3 else 0)      # last sweep: the sequential-sum path of every kernel
        assert rs.tokens == c.total_tokens == ro["stats"]["tokens"]
        assert rs.changed == ro["stats"]["changed"]
        assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("prio", ["0", "1", "2"])
def test_side_stream_priorities_never_change_a_result(prio, monkeypatch):
    """Every kernel class runs on a stream of its own, the widest on high-priority streams (a hardware-queue pool of their own,
    mvhdp_plan.h; MVHDP_SIDE_PRIORITY: 0 none, 1 the 4-round class too, 2 the default): where a kernel is queued is never what it computes."""
    monkeypatch.setenv("MVHDP_SIDE_PRIORITY", prio)
    K, V = 2048, [5000, 300]
    rng = np.random.RandomState(22)
    lens0 = np.array([60, 110, 150, 260, 300, 520, 700, 1100, 1500, 9, 0, 33, 64, 128, 256, 512, 1024] + [20] * 40, dtype=np.int64)
    lens1 = rng.randint(0, 12, len(lens0)).astype(np.int64)
    off0 = np.concatenate([[0], np.cumsum(lens0)]); off1 = np.concatenate([[0], np.cumsum(lens1)])
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, [off0, off1], [rng.randint(0, 5000, off0[-1]).astype(np.int32), rng.randint(0, 300, off1[-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    for it in range(3):
        ro = o.sweep(it, 32)
        rs = s.sweep(it, 32)
        assert rs.tokens == c.total_tokens and rs.changed == ro["stats"]["changed"]
        assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("mode", ["serial", "streams"])
def test_views_longer_than_16_bit_counts_take_the_generic_kernel(mode, monkeypatch):
    """The 8- and 16-round variants count tokens per slot in 16 bits: an entity with a view of more than 65535 tokens
    is sent to the generic kernel instead (by the overflow chain, or by the classify pass even when this corpus has no
    topic list beyond 1024 slots)."""
    monkeypatch.setenv("MVHDP_FORCE_MODE", mode)
    K, V = 600, [2000, 40]
    rng = np.random.RandomState(3)
    lens0 = np.array([70000, 5, 0, 300, 66000], dtype=np.int64); lens1 = np.array([3, 0, 0, 7, 2], dtype=np.int64)
    off = [np.concatenate([[0], np.cumsum(l)]) for l in (lens0, lens1)]
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, off, [rng.randint(0, V[m], off[m][-1]).astype(np.int32) for m in range(2)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(2)])
    for it in range(2):
        ro = o.sweep(it, 5); rs = s.sweep(it, 5)
        assert rs.tokens == c.total_tokens == ro["stats"]["tokens"]
        assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("K,V,D,lam", [
    (1, [30], 20, [6]),                                  # a single topic: FTree of size 1 (no descent)
    (2, [30, 7], 30, [9, 3]),
    (63, [200], 40, [80]), (64, [200], 40, [80]), (65, [200], 40, [80]),   # around one slot round / one bitmap word pair
    (1000, [900, 60, 60, 60, 60], 24, [96, 7, 7, 7, 7]),  # C5 shape (5 views, K=1000): generic kernel territory
    (2048, [300], 12, [40]),                             # MVHDP_MAX_TOPICS
    (2048, [900] + [50] * 7, 10, [3000] + [5] * 7),      # both maxima at once (K=2048, 8 views), lists beyond 1024 slots:
                                                         #   the n_k deltas no longer fit LDS and go straight to the delta buffer
    (1500, [900] + [50] * 5, 10, [500] + [5] * 5),
    (7, [900] + [50] * 7, 10, [300] + [5] * 7),
])
def test_edge_shapes(K, V, D, lam):
    c = small_corpus(K, V, D, lam, 1000 + K)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    for it in range(2):
        ro = o.sweep(it, 99); rs = s.sweep(it, 99)
        assert rs.tokens == ro["stats"]["tokens"] and rs.changed == ro["stats"]["changed"]
        assert_same_state(o, s, c.M)
    s.close()


def test_eight_views():
    K, V = 40, [300, 20, 20, 20, 20, 20, 20, 20]
    c = small_corpus(K, V, 30, [30, 3, 3, 3, 3, 3, 3, 3], 77)
    hy = Hyper.defaults(K, V)
    hy.gamma[:] = np.linspace(0.5, 1.5, 8)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(8)])
    for it in range(2):
        o.sweep(it, 3); s.sweep(it, 3)
        assert_same_state(o, s, 8)
    s.close()


@pytest.mark.parametrize("mode,force", [("serial", "1"), ("serial", "16"), ("streams", "1"), ("streams", "4"), ("streams", "16")])
def test_inactive_topic_activation_across_kernel_classes(mode, force, monkeypatch):
    """The truncated-HDP branch with entities spread over several kernel classes: the first delta on an inactive
    topic (in entity, view, position order, UPD:263-270) is found by atomicMin over every kernel of the sweep."""
    monkeypatch.setenv("MVHDP_FORCE_MODE", mode); monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    K, V = 700, [3000, 60]
    rng = np.random.RandomState(8)
    lens0 = np.array([900, 40, 1500, 10, 300, 80, 2500] + [25] * 30, dtype=np.int64)
    lens1 = rng.randint(0, 9, len(lens0)).astype(np.int64)
    off = [np.concatenate([[0], np.cumsum(l)]) for l in (lens0, lens1)]
    from mvtopicmodel_amd.synth import Corpus
    c = Corpus(K, V, off, [rng.randint(0, V[m], off[m][-1]).astype(np.int32) for m in range(2)])
    inactive = np.zeros(K, dtype=np.uint8); inactive[[650, 660, 699]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive); hy.alpha[:, K] = 40.0
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    for m in range(2):
        z0[m][np.isin(z0[m], [650, 660, 699])] = 2
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    activated = []
    for it in range(3):
        ro = o.sweep(it, 13); rs = s.sweep(it, 13)
        assert (ro["stats"]["activated_topic"], ro["stats"]["activated_modality"]) == (rs.activated_topic, rs.activated_modality)
        assert ro["stats"]["new_mass_cnt"] == rs.new_mass_cnt
        activated.append(rs.activated_topic)
        assert_same_state(o, s, 2)
    assert activated[0] == 650
    s.close()


def test_no_entities_and_a_view_without_tokens():
    from mvtopicmodel_amd import NativeSampler
    from mvtopicmodel_amd.synth import Corpus
    s = NativeSampler(10, [20, 5])
    for m in range(2):
        s.set_corpus(m, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32)); s.set_assignments(m, np.zeros(0, dtype=np.int32))
    s.set_hyper(Hyper.defaults(10, [20, 5])); s.build_counts()
    assert s.sweep(0, 1).tokens == 0
    s.close()
    c = small_corpus(30, [100, 20], 25, [20, 3], 5)
    c2 = Corpus(30, [100, 20], [c.doc_off[0], np.zeros(c.D + 1, dtype=np.int64)], [c.tokens[0], np.zeros(0, dtype=np.int32)])
    hy = Hyper.defaults(30, [100, 20])
    o = make_oracle(c2, hy); s = make_native(c2, hy, [o.get_assignments(m) for m in range(2)])
    for it in range(2):
        o.sweep(it, 2); s.sweep(it, 2); assert_same_state(o, s, 2)
    s.close()


@pytest.mark.parametrize("flags_name", ["deferred", "live", "segmented", "frozen"])
def test_sweep_many_equals_single_sweeps(flags_name):
    """mvhdp_sweep_many: n sweeps enqueued back to back under ONE plan, no host round trip in between -- the integers of n single
    calls (and so the oracle's), statistics per sweep included.  With inactive topics it falls back to single calls (the activation
    needs the host): same results again."""
    from mvtopicmodel_amd.native import SWEEP_FROZEN, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_SEGMENT_APPLY
    from tests.test_gpu_segmented import oracle_segmented_sweep
    K, V = 90, [900, 120, 100]
    c = small_corpus(K, V, 260, [70, 6, 8], 71)
    for inactive in (None, np.eye(1, K, 83, dtype=np.uint8)[0]):
        hy = Hyper.defaults(K, V, inactive=inactive)
        if inactive is not None:
            hy.alpha[:, K] = 20.0
        o = make_oracle(c, hy)
        z0 = [o.get_assignments(m) for m in range(3)]
        for m in range(3):
            z0[m][z0[m] == 83] = 1
            o.set_assignments(m, z0[m])
        o.build_counts()
        s = make_native(c, hy, z0)
        n = 6
        if flags_name == "deferred":
            sts = s.sweep_many(3, n, 99)
            for i in range(n):
                ro = o.sweep(3 + i, 99)
                assert (sts[i].tokens, sts[i].changed, sts[i].word_ftree_mass_cnt) == (ro["stats"]["tokens"], ro["stats"]["changed"], ro["stats"]["word_ftree_mass_cnt"])
                if inactive is not None:
                    assert sts[i].activated_topic == ro["stats"]["activated_topic"]
            assert_same_state(o, s, 3)
        elif flags_name == "segmented":
            sts = s.sweep_many(3, n, 99, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(3))
            for i in range(n):
                so, _ = oracle_segmented_sweep(o, c, 3 + i, 99, 3)
                assert (sts[i].tokens, sts[i].changed) == (so["tokens"], so["changed"])
            assert_same_state(o, s, 3)
        elif flags_name == "frozen":
            s.build_trees(); o.build_trees()
            before = [s.get_counts(m) for m in range(3)]
            from oracle.binding import SWEEP_FROZEN as ORC_FROZEN
            sts = s.sweep_many(3, n, 99, flags=SWEEP_FROZEN)
            for i in range(n):
                o.sweep(3 + i, 99, flags=ORC_FROZEN)
            for m in range(3):
                assert np.array_equal(o.get_assignments(m), s.get_assignments(m))
                assert np.array_equal(before[m][0], s.get_counts(m)[0])           # the model is untouched
        else:
            sts = s.sweep_many(3, n, 99, flags=SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(2))
            assert all(st.tokens == c.total_tokens for st in sts)
            for m in range(3):
                z = s.get_assignments(m); nwk, nk = s.get_counts(m)
                ref = np.zeros_like(nwk); np.add.at(ref, (c.tokens[m], z), 1)
                assert nwk.min() >= 0 and np.array_equal(ref, nwk) and np.array_equal(ref.sum(axis=0), nk)
        s.close()
