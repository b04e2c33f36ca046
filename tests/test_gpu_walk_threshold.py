"""The thresholded tree walk of the chunk head (SweepLaunch::walk_theta, mvhdp_sweep_fast.hip): a token whose first
uniform is below the view's threshold is not walked up front and, if it reaches the tree branch (WRK:533-535) after all,
walks its word's tree on demand inside the token loop.  The threshold decides WHEN FTree.sample (FT:111-136) is
evaluated, never what it returns: whatever the thresholds, assignments, counts, branch counters and masses must be the
oracle's, bit for bit.  MVHDP_WALK_THETA fixes the thresholds (a diagnostic switch read by every mvhdp_sweep call);
without it the library searches for them by the clock, which the last test lets run."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper, SWEEP_EXACT_CHAIN
from tests.helpers import assert_same_state, make_native, make_oracle, small_corpus

pytestmark = pytest.mark.gpu

CASES = [
    (20, [300, 40, 50], 64, [30, 4, 6], 12),
    (100, [2000], 96, [127], 13),
    (200, [3000, 300, 300], 80, [127, 7, 15], 15),        # 8 levels: blocks of 3 + 3 + 2
    (400, [5000, 500, 500], 48, [127, 7, 15], 16),        # 9 levels, 2-round variant
    (1000, [3000, 200, 200, 200, 200], 24, [600, 20, 20, 20, 20], 17),   # wide variants (LDS slot counts)
]


@pytest.mark.parametrize("theta", ["0", "0.5", "0.85", "1.1"])
@pytest.mark.parametrize("K,V,D,lam,cseed", CASES)
def test_any_threshold_gives_the_oracles_sweep(monkeypatch, K, V, D, lam, cseed, theta):
    """theta 0: every token walked up front; 0.5 / 0.85: a mix; 1.1: no token walked up front, every tree-branch
    token walks on demand."""
    monkeypatch.setenv("MVHDP_WALK_THETA", ",".join([theta] * len(V)))
    c = small_corpus(K, V, D, lam, cseed)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    s = make_native(c, hy, z0)
    for it in range(2):
        ro = o.sweep(it, 0xBEEF, want_dbg=(it == 1))
        rs = s.sweep(it, 0xBEEF, want_dbg=(it == 1))
        for f in ("tokens", "changed", "new_mass_cnt", "topic_doc_mass_cnt", "word_ftree_mass_cnt", "oov_skipped", "aborted_docs"):
            assert ro["stats"][f] == getattr(rs, f), f
        assert rs.word_ftree_mass_cnt > 0
        assert_same_state(o, s, c.M)
    for m in range(c.M):
        a, b = ro["dbg"][m], rs.dbg[m]
        assert np.array_equal(a[:, 0], b[:, 0])
        assert np.array_equal(a[:, 2], b[:, 2])              # tree[1]: from the walked block or from the root array, same double
        assert np.allclose(a[:, 1], b[:, 1], rtol=1e-12, atol=0)
    s.close()


@pytest.mark.parametrize("theta", ["0.6", "1.1"])
def test_threshold_with_inactive_topics_oov_and_exact_chain(monkeypatch, theta):
    """The truncated-HDP branch (new topic, WRK:515-526), out-of-vocabulary types and the sequential-sum mode under a threshold."""
    monkeypatch.setenv("MVHDP_WALK_THETA", f"{theta},{theta}")
    K, V = 40, [400, 60]
    c = small_corpus(K, V, 64, [50, 6], 21)
    oov = np.arange(0, c.tokens[0].size, 17)
    c.tokens[0][oov] = V[0] + 5                                  # OOV types (WRK:427-428)
    inactive = np.zeros(K, dtype=np.uint8); inactive[[33, 36, 39]] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    hy.alpha[:, K] = 25.0                                        # make the new-topic branch likely
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(2)]
    for m in range(2):                                           # inactive topics hold no tokens, OOV tokens no topic
        z0[m][np.isin(z0[m], [33, 36, 39])] = 1
    z0[0][oov] = -1
    for m in range(2):
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    s2 = make_native(c, hy, z0)
    for it in range(3):
        ro = o.sweep(it, 7)
        rs = s.sweep(it, 7)
        s2.sweep(it, 7, flags=SWEEP_EXACT_CHAIN)
        assert ro["stats"]["new_mass_cnt"] == rs.new_mass_cnt and ro["stats"]["oov_skipped"] == rs.oov_skipped > 0
        assert_same_state(o, s, c.M)
        assert_same_state(o, s2, c.M)
    s.close(); s2.close()


def test_threshold_search_never_changes_the_chain():
    """The library's own search (no MVHDP_WALK_THETA): over 14 sweeps it runs base and probe sweeps at different
    thresholds, in both kernel flavours; the chain stays the oracle's."""
    K, V = 100, [1500, 200]
    c = small_corpus(K, V, 200, [100, 8], 31)
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    s = make_native(c, hy, z0)
    for it in range(14):
        o.sweep(it, 99)
        s.sweep(it, 99)
        assert_same_state(o, s, c.M)
    s.close()


@pytest.mark.parametrize("narrow", ["1", "0"])
def test_16_bit_mirror_of_the_counts_including_saturated_cells(monkeypatch, narrow):
    """The 1-round walk flavour gathers n_wk from the 16-bit mirror the tree build writes (MvModel::counts16); a cell
    that does not fit 16 bits reads 65535 there and is fetched from the 32-bit table.  Two types and 600 k tokens put
    most cells far beyond 65535; the sweep must still be the oracle's (and the same with the mirror switched off)."""
    monkeypatch.setenv("MVHDP_WALK_THETA", "0.5")           # the walk flavour (a threshold of 0 would run the plain one)
    monkeypatch.setenv("MVHDP_NARROW", narrow)
    from mvtopicmodel_amd.synth import Corpus
    K, V, D = 6, [2], 3000
    rng = np.random.RandomState(5)
    lens = np.full(D, 200, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    tok = (rng.rand(off[-1]) < 0.7).astype(np.int32)
    c = Corpus(K, V, [off], [tok])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(0)]
    s = make_native(c, hy, z0)
    assert s.get_counts(0)[0].max() > 65535
    for it in range(2):
        ro = o.sweep(it, 4)
        rs = s.sweep(it, 4)
        assert ro["stats"]["word_ftree_mass_cnt"] == rs.word_ftree_mass_cnt and ro["stats"]["changed"] == rs.changed
        assert_same_state(o, s, 1)
    s.close()


@pytest.mark.parametrize("force", ["", "2", "4"])
@pytest.mark.parametrize("delta16", ["1", "0"])
def test_16_bit_delta_cells_with_the_three_row_classes(monkeypatch, force, delta16):
    """A plain deferred sweep of the narrow flavour keeps the n_wk deltas of the rows that cannot overflow them in 16-bit cells biased by
    0x8000 (MvModel::delta16, SweepLaunch::delta16: half the table its chunk-end atomics land in); the apply pass adds them to the 32-bit
    deltas and re-biases the cells.  One corpus with the three row classes build_trees_kernel hands out -- a type with more than 65534
    tokens (HEAVY: 32-bit counts and deltas), one between 32768 and 65534 (mirror counts, 32-bit deltas), the rest small (mirror counts,
    16-bit deltas) -- swept five times: every state is the oracle's, with the 16-bit cells and without."""
    monkeypatch.setenv("MVHDP_WALK_THETA", "0.3")           # the walk flavour, hence the mirror and the row classes
    monkeypatch.setenv("MVHDP_DELTA16", delta16)
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    from mvtopicmodel_amd.synth import Corpus
    K, V, D = 24, [40, 7], 2500
    rng = np.random.RandomState(12)
    lens0 = np.full(D, 200, dtype=np.int64); lens1 = rng.randint(0, 5, D).astype(np.int64)
    off = [np.concatenate([[0], np.cumsum(l)]) for l in (lens0, lens1)]
    u = rng.rand(off[0][-1])
    t0 = rng.randint(2, 40, off[0][-1]).astype(np.int32)
    t0[u < 0.5] = 0                                                    # type 0: 250 k tokens
    t0[(u >= 0.5) & (u < 0.59)] = 1                                    # type 1: 45 k
    c = Corpus(K, V, off, [t0, rng.randint(0, 7, off[1][-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    tot = s.get_counts(0)[0].sum(axis=1)
    assert tot[0] > 65534 and 32767 < tot[1] <= 65534 and tot[2:].max() <= 32767
    for it in range(5):
        ro = o.sweep(it, 33)
        rs = s.sweep(it, 33)
        assert ro["stats"]["changed"] == rs.changed and rs.tokens == c.total_tokens
        assert_same_state(o, s, c.M)
    # a sweep that leaves its deltas for the host (document shards) keeps them all in the 32-bit table
    from mvtopicmodel_amd.native import SWEEP_NO_APPLY
    o.sweep(5, 33); s.sweep(5, 33, flags=SWEEP_NO_APPLY); s.apply_delta(-1, -1)
    assert_same_state(o, s, c.M)
    # ... and batches (mvhdp_sweep_many) fold the cells in after every sweep of the batch
    for it in (6, 7, 8):
        o.sweep(it, 33)
    s.sweep_many(6, 3, 33)
    assert_same_state(o, s, c.M)
    s.close()
