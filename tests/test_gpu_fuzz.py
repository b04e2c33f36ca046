"""Seeded random shapes against the oracle: topics, views, vocabulary sizes, entity lengths (with empty views and
entities), unassigned tokens, inactive topics, out-of-vocabulary types, primary kernel variant and dispatch mode all
drawn at random; integers must agree after every sweep."""
import os

import numpy as np
import pytest

from tests.helpers import assert_same_state, make_native, make_oracle
from mvtopicmodel_amd.native import Hyper, SWEEP_EXACT_CHAIN, SWEEP_GENERIC_KERNEL, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS
from mvtopicmodel_amd.synth import Corpus

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.RandomState(seed)
    M = int(rng.choice([1, 1, 2, 3, 3, 5, 8]))
    K = int(rng.choice([1, 2, 3, 17, 64, 65, 100, 129, 300, 511, 700, 1200, 2048]))
    V = [int(rng.randint(1, 3000))] + [int(rng.randint(1, 120)) for _ in range(M - 1)]
    D = int(rng.randint(1, 40))
    scale = [int(rng.choice([3, 30, 200, 900]))] + [int(rng.choice([1, 4, 12])) for _ in range(M - 1)]
    offs, toks = [], []
    for m in range(M):
        lens = rng.poisson(scale[m], D).astype(np.int64)
        lens[rng.rand(D) < 0.15] = 0                                  # entities that lack this view
        if m == 0 and rng.rand() < 0.5:
            lens[rng.randint(D)] = int(rng.choice([1500, 2600]))      # one long entity: wide variants / generic kernel
        off = np.concatenate([[0], np.cumsum(lens)])
        tk = rng.randint(0, V[m], off[-1]).astype(np.int32)
        if len(tk) and rng.rand() < 0.2:
            tk[rng.rand(len(tk)) < 0.05] = V[m] + int(rng.randint(0, 5))   # types outside the vocabulary (WRK:427-428)
        offs.append(off); toks.append(tk)
    c = Corpus(K, V, offs, toks)
    inactive = None
    if K >= 17 and rng.rand() < 0.4:
        inactive = np.zeros(K, dtype=np.uint8); inactive[rng.choice(K, size=min(3, K - 1), replace=False)] = 1
    hy = Hyper.defaults(K, V, inactive=inactive)
    if inactive is not None:
        hy.alpha[:, K] = float(rng.choice([0.5, 8.0, 40.0]))
    hy.gamma[:] = rng.uniform(0.3, 2.0, M)
    hy.beta[:] = rng.choice([0.01, 0.05, 0.2], M); hy.beta_sum[:] = hy.beta * np.array(V)
    flags = int(rng.choice([0, 0, 0, SWEEP_EXACT_CHAIN, SWEEP_GENERIC_KERNEL]))
    # the segmented launch path (interleaved queue segments) under the deferred contract: same integers for any count
    flags |= SWEEP_LIVE_SEGMENTS(int(rng.choice([0, 0, 1, 2, 3, 7, 40])))
    env = {"MVHDP_FORCE_MODE": str(rng.choice(["", "serial", "streams"])),
           "MVHDP_FORCE_RMAX": str(rng.choice(["", "", "1", "2", "4", "8", "16"]))}
    # the walk threshold of the chunk head (when a token's word tree is walked, never what it returns): left to the library's
    # search, or pinned -- 0 = every token up front, > 1 = every tree-branch token on demand, or a random one per view
    walk = str(rng.choice(["", "", "0", "1.1", "rand"]))             # (drawn last: the shapes of earlier rounds' cases stay)
    if walk == "rand":
        walk = ",".join("%.2f" % t for t in rng.uniform(0.0, 1.0, M))
    env["MVHDP_WALK_THETA"] = walk
    return c, hy, inactive, flags, env, rng


@pytest.mark.parametrize("seed", range(int(os.environ.get("MVHDP_FUZZ_CASES", "40"))))
def test_random_shapes(seed, monkeypatch):
    c, hy, inactive, flags, env, rng = _case(1000 + seed)
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        if inactive is not None:
            z0[m][np.isin(z0[m], np.flatnonzero(inactive))] = int(np.flatnonzero(inactive == 0)[0])
        if len(z0[m]) and rng.rand() < 0.3:
            z0[m][rng.rand(len(z0[m])) < 0.1] = -1                    # UNASSIGNED_TOPIC PTM:63
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    for it in range(3):
        ro = o.sweep(it, 77 + seed); rs = s.sweep(it, 77 + seed, flags=flags)
        st = ro["stats"]
        assert (rs.tokens, rs.changed, rs.new_mass_cnt, rs.topic_doc_mass_cnt, rs.word_ftree_mass_cnt, rs.aborted_docs, rs.oov_skipped) == \
               (st["tokens"], st["changed"], st["new_mass_cnt"], st["topic_doc_mass_cnt"], st["word_ftree_mass_cnt"], st["aborted_docs"], st["oov_skipped"])
        assert (rs.activated_topic, rs.activated_modality) == (st["activated_topic"], st["activated_modality"])
        assert_same_state(o, s, c.M)
    s.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MVHDP_FUZZ_CASES", "40")) // 2))
def test_random_shapes_statistics_and_shards(seed):
    """The statistics either side of the sweep (countHistogram, topic/length histograms, view overlap sums, log-likelihood,
    topic proportions) and the document-shard identity on the same kind of random shapes."""
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.native import SWEEP_NO_APPLY
    from oracle import doc_topics as dto
    c, hy, inactive, flags, env, rng = _case(5000 + seed)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    if inactive is not None:
        for m in range(c.M):
            z0[m][np.isin(z0[m], np.flatnonzero(inactive))] = int(np.flatnonzero(inactive == 0)[0])
            o.set_assignments(m, z0[m])
        o.build_counts()
    s = make_native(c, hy, z0)
    o.sweep(0, 9); s.sweep(0, 9)
    M, K = c.M, c.K
    for m in range(M):
        mx = int(max(o.get_counts(m)[0].max(), 1))
        assert np.array_equal(o.count_histogram(m, mx + 1), s.get_count_histogram(m, mx + 1))
        hl = int(np.diff(c.doc_off[m]).max()) + 1
        ho, lo_ = o.get_doc_topic_hist(m, hl, hl); hs, ls_ = s.get_doc_topic_hist(m, hl, hl)
        assert np.array_equal(ho, hs) and np.array_equal(lo_, ls_)
    if M > 1:
        assert np.array_equal(o.optimize_p_sums(), s.view_overlap_sums())
    llo, lls = o.model_log_likelihood(), s.model_log_likelihood()
    assert np.allclose(llo, lls, rtol=1e-11, atol=1e-9), (llo, lls)
    w = rng.uniform(0.2, 1.0, M)
    z = [s.get_assignments(m) for m in range(M)]
    if all((zz >= 0).all() for zz in z):
        alpha_now = s.get_alpha()[0]                  # a topic activated by the sweep took over alpha[m][K] (UPD:268)
        want = dto.doc_topic_proportions(K, c.doc_off, z, alpha_now, hy.alpha_sum, hy.gamma, w)
        assert np.array_equal(s.doc_topic_proportions(w), want)

    # two or three document shards on the same GPU, deltas summed on the host, against the single sampler
    n_sh = int(rng.choice([2, 3]))
    if c.D >= n_sh:
        tot = sum(np.diff(c.doc_off[m]) for m in range(M))
        bounds = synth.shard_bounds(tot, n_sh)
        glob = [s.get_counts(m) for m in range(M)]
        shards = []
        for lo, hi in bounds:
            sub = c.slice_docs(lo, hi)
            sh = make_native(sub, hy, [z[m][c.doc_off[m][lo]:c.doc_off[m][hi]] for m in range(M)], doc_id_base=lo)
            for m in range(M):
                sh.set_counts(m, *glob[m])
            shards.append(sh)
        if inactive is not None:                      # replicas must hold the single sampler's alpha / inactive set too
            a, ina = s.get_alpha()
            hy2 = Hyper.defaults(K, c.V, inactive=ina); hy2.alpha[:] = a; hy2.alpha_sum[:] = hy.alpha_sum
            hy2.gamma[:] = hy.gamma; hy2.beta[:] = hy.beta; hy2.beta_sum[:] = hy.beta_sum
            for sh in shards:
                sh.set_hyper(hy2)
        st1 = s.sweep(1, 9, flags=SWEEP_NO_APPLY)
        sts = [sh.sweep(1, 9, flags=SWEEP_NO_APPLY) for sh in shards]
        assert sum(x.tokens for x in sts) == st1.tokens and sum(x.changed for x in sts) == st1.changed
        keys = [x.activation_key for x in sts if x.activated_topic >= 0]
        assert (min(keys) if keys else None) == (st1.activation_key if st1.activated_topic >= 0 else None)
        for m in range(M):
            assert np.array_equal(np.concatenate([sh.get_assignments(m) for sh in shards]), s.get_assignments(m))
        for sh in shards:
            sh.close()
    s.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MVHDP_FUZZ_CASES", "40")) // 2))
def test_random_shapes_live(seed, monkeypatch):
    """MVHDP_SWEEP_LIVE on the same random shapes (variants, dispatch modes, unassigned and out-of-vocabulary tokens,
    inactive topics): whatever the interleaving, the counts are the counts of z and the statistics add up."""
    c, hy, inactive, flags, env, rng = _case(9000 + seed)
    env["MVHDP_LIVE16"] = str(seed % 2)                   # even: atomics and gathers on the 32-bit table; odd: light rows live in the 16-bit mirror
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        if inactive is not None:
            z0[m][np.isin(z0[m], np.flatnonzero(inactive))] = int(np.flatnonzero(inactive == 0)[0])
        if len(z0[m]) and rng.rand() < 0.3:
            z0[m][rng.rand(len(z0[m])) < 0.1] = -1
    s = make_native(c, hy, z0)
    tokens_in_vocab = sum(int(((c.tokens[m] >= 0) & (c.tokens[m] < c.V[m])).sum()) for m in range(c.M))
    for it in range(3):
        st = s.sweep(it, 5 + seed, flags=(flags & ~0xFF0000) | SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(int(rng.choice([0, 1, 2, 5]))))
        assert st.tokens == tokens_in_vocab and st.aborted_docs == 0
        assert st.new_mass_cnt + st.topic_doc_mass_cnt + st.word_ftree_mass_cnt == st.tokens
        for m in range(c.M):
            z = s.get_assignments(m)
            nwk, nk = s.get_counts(m)
            ok = (z >= 0) & (c.tokens[m] >= 0) & (c.tokens[m] < c.V[m])
            ref = np.zeros_like(nwk); np.add.at(ref, (c.tokens[m][ok], z[ok]), 1)
            assert nwk.min() >= 0 and np.array_equal(ref, nwk) and np.array_equal(ref.sum(axis=0), nk)
    s.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MVHDP_FUZZ_CASES", "40")) // 2))
def test_random_shapes_segmented(seed, monkeypatch):
    """MVHDP_SWEEP_SEGMENT_APPLY on random shapes: deterministic, so every integer must equal the oracle's, which follows
    the sweep segment by segment (tests/test_gpu_segmented.py)."""
    from mvtopicmodel_amd.native import SWEEP_SEGMENT_APPLY
    from tests.test_gpu_segmented import oracle_segmented_sweep
    c, hy, inactive, flags, env, rng = _case(13000 + seed)
    for k, v in env.items():
        if v:
            monkeypatch.setenv(k, v)
    o = make_oracle(c, hy)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    for m in range(c.M):
        if inactive is not None:
            z0[m][np.isin(z0[m], np.flatnonzero(inactive))] = int(np.flatnonzero(inactive == 0)[0])
        if len(z0[m]) and rng.rand() < 0.3:
            z0[m][rng.rand(len(z0[m])) < 0.1] = -1
        o.set_assignments(m, z0[m])
    o.build_counts()
    s = make_native(c, hy, z0)
    nseg = int(rng.choice([2, 3, 4, 9]))
    for it in range(3):
        so, best = oracle_segmented_sweep(o, c, it, 31 + seed, nseg)
        st = s.sweep(it, 31 + seed, flags=(flags & 0xFFFF) | SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(nseg))
        assert (st.tokens, st.changed, st.new_mass_cnt, st.topic_doc_mass_cnt, st.word_ftree_mass_cnt) == \
               (so["tokens"], so["changed"], so["new_mass_cnt"], so["topic_doc_mass_cnt"], so["word_ftree_mass_cnt"])
        assert (st.activated_topic, st.activated_modality, st.activations) == (best[1], best[2], best[3])
        assert_same_state(o, s, c.M)
        if inactive is not None:
            assert np.array_equal(s.get_alpha()[0], o.get_alpha()) and np.array_equal(s.get_alpha()[1], o.get_inactive())
    s.close()


@pytest.mark.parametrize("seed", range(max(4, int(os.environ.get("MVHDP_FUZZ_CASES", "40")) // 10)))
def test_random_row_classes(seed, monkeypatch):
    """Types with 15 k ... 100 k tokens in one corpus: the three row classes of build_trees_kernel (16-bit deltas and mirror counts,
    32-bit deltas and mirror counts, 32-bit everything) in random proportions, under a random walk threshold (the narrow flavour),
    a random primary variant and, for one sweep in three, as a batch or with the deltas left to the host."""
    rng = np.random.RandomState(5000 + seed)
    K = int(rng.choice([8, 24, 64, 130]))
    V = [int(rng.randint(3, 40)), int(rng.randint(1, 9))]
    total = int(rng.choice([50_000, 90_000, 130_000]))
    D = int(rng.randint(40, 400))
    lens0 = rng.multinomial(total, np.ones(D) / D).astype(np.int64)
    lens1 = rng.randint(0, 6, D).astype(np.int64)
    off = [np.concatenate([[0], np.cumsum(l)]) for l in (lens0, lens1)]
    share = rng.dirichlet(np.full(V[0], float(rng.choice([0.05, 0.3, 1.0]))))          # a few dominant types
    t0 = rng.choice(V[0], size=off[0][-1], p=share).astype(np.int32)
    c = Corpus(K, V, off, [t0, rng.randint(0, V[1], off[1][-1]).astype(np.int32)])
    hy = Hyper.defaults(K, V)
    hy.gamma[:] = rng.uniform(0.3, 2.0, 2)
    monkeypatch.setenv("MVHDP_WALK_THETA", "%.2f,%.2f" % tuple(rng.uniform(0.05, 0.9, 2)))
    force = str(rng.choice(["", "", "1", "2", "4"]))
    if force:
        monkeypatch.setenv("MVHDP_FORCE_RMAX", force)
    monkeypatch.setenv("MVHDP_DELTA16", str(rng.choice(["1", "1", "1", "0"])))
    o = make_oracle(c, hy)
    s = make_native(c, hy, [o.get_assignments(m) for m in range(c.M)])
    from mvtopicmodel_amd.native import SWEEP_NO_APPLY
    it = 0
    for step in range(3):
        kind = int(rng.choice([0, 0, 1, 2]))
        if kind == 1:                                                   # a batch of two sweeps
            o.sweep(it, 91 + seed); o.sweep(it + 1, 91 + seed)
            s.sweep_many(it, 2, 91 + seed); it += 2
        elif kind == 2:                                                 # deltas left to the host, applied by it
            o.sweep(it, 91 + seed); s.sweep(it, 91 + seed, flags=SWEEP_NO_APPLY); s.apply_delta(-1, -1); it += 1
        else:
            ro = o.sweep(it, 91 + seed); rs = s.sweep(it, 91 + seed); it += 1
            assert rs.changed == ro["stats"]["changed"] and rs.tokens == c.total_tokens
        assert_same_state(o, s, c.M)
    s.close()
