"""Every BASELINE config at its stated size on one GPU: C2 (50 k x 1 view, K=100), C3 (200 k x 3, K=200),
C4 (1 M x 3, K=400) and C5 (1 M x 5, K=1000, power-law lengths, the top 10 % of the topic ids inactive).

No oracle can follow a whole corpus of that size, so each config is checked three ways:
  * properties that need no oracle: every count the reference keeps is conserved (PTM:511,640-643,872), n_wk is exactly
    the recount of z (numpy bincount), the sweep statistics add up (WRK:33-35);
  * the oracle on a PREFIX: under the snapshot contract entities are independent, so the first 2000 entities sampled by
    the oracle against the same global counts must get the very same assignments, bit for bit, in both sweeps;
  * shard independence: the same two sweeps on two document shards (token-balanced cut, global entity ids, deltas
    summed as RCCL's all_reduce(SUM) would, activation key MIN-reduced) give the same integers and, for C5, activate
    the same topic in the same view."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper

pytestmark = pytest.mark.gpu

PREFIX = 2000


def _fingerprint(s, M):
    out = []
    for m in range(M):
        nwk, nk = s.get_counts(m)
        w = (np.arange(nk.size, dtype=np.int64) % 977) + 1
        out.append(int((nk.astype(np.int64) * w).sum()))
        out.append(int((nwk.astype(np.int64).sum(axis=0) * w).sum()))
    return out


def _recount(tokens, z, V, K):
    return np.bincount(tokens.astype(np.int64) * K + z, minlength=V * K).reshape(V, K)


def _oracle_prefix(c, hy_arrays, z_prefix, counts, lo=0, hi=PREFIX):
    from oracle.binding import Oracle
    sub = c.slice_docs(lo, hi)
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, sub.doc_off[m], sub.tokens[m])
        o.set_assignments(m, z_prefix[m])
        o.set_counts(m, *counts[m])
    alpha, inactive, hy = hy_arrays
    o.set_hyper(alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, inactive)
    return o


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_config_at_full_size(name):
    import torch
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.dist import KEY_NONE, GpuShard, decode_activation
    from mvtopicmodel_amd.java_init import init_assignments
    cfg = synth.CONFIGS[name]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(name)
    assert c.D == cfg["D"]
    inactive, K_init = synth.config_inactive(name)
    z0 = init_assignments(K_init, c.doc_off, seed=1)                  # PTM:465-515 with java.util.Random(1)
    hy = Hyper.defaults(K, V, inactive=inactive)
    type_totals = [np.bincount(c.tokens[m], minlength=V[m]).astype(np.int64) for m in range(M)]
    pre_off = [int(c.doc_off[m][PREFIX]) for m in range(M)]

    one = NativeSampler(K, V)
    for m in range(M):
        one.set_corpus(m, c.doc_off[m], c.tokens[m]); one.set_assignments(m, z0[m])
    one.set_hyper(hy); one.build_counts()
    z_prev = [z0[m][:pre_off[m]].copy() for m in range(M)]
    acts = []
    for it in range(2):
        counts_before = [one.get_counts(m) for m in range(M)]
        if it == 0:                                                    # PTM:600-652 against an independent recount
            for m in range(M):
                assert np.array_equal(counts_before[m][0], _recount(c.tokens[m], z0[m], V[m], K))
        alpha_before, inactive_before = one.get_alpha()
        st = one.sweep(it, 1)
        assert st.tokens == c.total_tokens and st.aborted_docs == 0 and st.oov_skipped == 0
        assert st.new_mass_cnt + st.topic_doc_mass_cnt + st.word_ftree_mass_cnt == st.tokens     # WRK:33-35
        assert 0.3 * st.tokens < st.changed <= st.tokens
        if inactive is not None:
            assert st.new_mass_cnt > 0 and st.activated_topic == int(np.flatnonzero(inactive_before)[0])   # WRK:515-526, UPD:263-270
        else:
            assert st.new_mass_cnt == 0 and st.activated_topic == -1
        acts.append((st.activated_topic, st.activated_modality, st.activation_key))
        if M > 1 and it == 0:
            # WRK:327-337 for EVERY entity: the device's log / pow / cos feed Math.round(1000 x) -- one flipped rounding in a
            # million entities would go unseen by the prefix check below
            from oracle.binding import Oracle
            ow = Oracle(K, V)
            for m in range(M):
                ow.set_corpus(m, np.zeros(c.D + 1, dtype=np.int64), np.zeros(0, dtype=np.int32))
            ow.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
            assert np.array_equal(ow.draw_p_philox(1, it), one.get_view_weights())
            ow.close()
        # the oracle follows the first PREFIX entities against the same snapshot
        o = _oracle_prefix(c, (alpha_before, inactive_before if inactive is not None else None, hy), z_prev, counts_before)
        o.sweep(it, 1, flags=ORC_NO_APPLY, doc_id_base=0)
        for m in range(M):
            zg = one.get_assignments(m)[:pre_off[m]]
            zo = o.get_assignments(m)
            assert np.array_equal(zg, zo), f"{name} sweep {it} view {m}: {np.count_nonzero(zg != zo)} of {len(zo)} prefix assignments differ from the oracle"
            z_prev[m] = zo
        o.close()
    z_one = [one.get_assignments(m) for m in range(M)]
    for m in range(M):
        nwk, nk = one.get_counts(m)
        assert nwk.min() >= 0
        assert np.array_equal(nwk.astype(np.int64).sum(axis=1), type_totals[m])               # PTM:872 typeTotals
        assert np.array_equal(nwk.astype(np.int64).sum(axis=0), nk.astype(np.int64))           # PTM:640-643
        assert int(nk.astype(np.int64).sum()) == int(c.doc_off[m][-1])                          # PTM:511 totalTokens
        assert z_one[m].min() >= 0 and z_one[m].max() < K
        assert np.array_equal(nwk, _recount(c.tokens[m], z_one[m], V[m], K))                    # n_wk is the count of z
    want = _fingerprint(one, M)
    alpha_one = one.get_alpha()
    one.close()

    # the same two sweeps on two document shards
    tot = sum(np.diff(c.doc_off[m]) for m in range(M))
    bounds = synth.shard_bounds(tot, 2)
    shards = []
    for lo, hi in bounds:
        sub = c.slice_docs(lo, hi)
        s = NativeSampler(K, V, doc_id_base=lo)
        for m in range(M):
            s.set_corpus(m, sub.doc_off[m], sub.tokens[m])
            s.set_assignments(m, z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
        s.set_hyper(hy); s.build_counts()
        shards.append(s)
    gs = [GpuShard(s, "cuda:0") for s in shards]
    torch.cuda.synchronize()
    total = gs[0].counts + gs[1].counts                              # all-reduce of the initial counts
    for g in gs:
        g.counts.copy_(total)
    torch.cuda.synchronize()
    for g in gs:
        g.counts_written()
    for it in range(2):
        sts = [g.sweep_local(it, 1) for g in gs]
        total = gs[0].delta + gs[1].delta                            # what RCCL's all_reduce(SUM) does across ranks
        for g in gs:
            g.delta.copy_(total)
        torch.cuda.synchronize()
        key = min(st.activation_key for st in sts)                   # all_reduce(MIN) of the activation key
        topic, modality = decode_activation(key)
        assert (topic, modality) == acts[it][:2] and (key == acts[it][2] or (key == KEY_NONE and acts[it][0] == -1))
        for g in gs:
            g.apply(topic, modality)
    assert _fingerprint(shards[0], M) == want and _fingerprint(shards[1], M) == want
    for m in range(M):
        zcat = np.concatenate([s.get_assignments(m) for s in shards])
        assert np.array_equal(zcat, z_one[m])
    for s in shards:
        a = s.get_alpha()
        assert np.array_equal(a[0], alpha_one[0]) and np.array_equal(a[1], alpha_one[1])
    for g in gs:
        g.close()
    for s in shards:
        s.close()


def _prefix_oracle(c, hy, one, inactive):
    """An oracle over the first PREFIX entities holding the handle's current prefix assignments; the global counts and
    hyper-parameters are set before every sweep it follows (entities are independent under the snapshot)."""
    from oracle.binding import Oracle
    sub = c.slice_docs(0, PREFIX)
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, sub.doc_off[m], sub.tokens[m])
        o.set_assignments(m, one.get_assignments(m)[:int(c.doc_off[m][PREFIX])])
    return o


def _sync_oracle_model(o, one, hy, inactive, M):
    for m in range(M):
        o.set_counts(m, *one.get_counts(m))
    alpha, ina = one.get_alpha()
    o.set_hyper(alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, ina if inactive is not None else None)


def _assert_prefix(o, one, c, what):
    for m in range(c.M):
        zg = one.get_assignments(m)[:int(c.doc_off[m][PREFIX])]
        zo = o.get_assignments(m)
        assert np.array_equal(zg, zo), f"{what}, view {m}: {np.count_nonzero(zg != zo)} of {len(zo)} prefix assignments differ from the oracle"


@pytest.mark.parametrize("name", ["C4", "C5"])
def test_the_kernels_that_carry_the_number_at_full_size(name):
    """the kernel flavours of the bench window meet the oracle at full size, on a chain that has run for a while
    (16 sweeps, enqueued as ONE batch: mvhdp_sweep_many):
      (a) the 1-round variant on the 16-bit mirror with a walk threshold, the longer lists on their own class kernels beside it;
      (b) the segmented sweep (8 segments): every segment against the oracle on the prefix, the host driving the segments one by one
          (MVHDP_SWEEP_ONLY_SEGMENT) -- and the single-call MVHDP_SWEEP_SEGMENT_APPLY on a twin handle gives the very same integers;
      (c) the live sweep (light rows on the 16-bit mirror): the counts are exactly the recount of z, nothing negative, sums right."""
    from oracle.binding import SWEEP_NO_APPLY as ORC_NO_APPLY
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_ONLY_SEGMENT, SWEEP_SEGMENT_APPLY
    cfg = synth.CONFIGS[name]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(name)
    inactive, K_init = synth.config_inactive(name)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    hy = Hyper.defaults(K, V, inactive=inactive)
    one = NativeSampler(K, V)
    for m in range(M):
        one.set_corpus(m, c.doc_off[m], c.tokens[m]); one.set_assignments(m, z0[m])
    del z0
    one.set_hyper(hy); one.build_counts()
    sts = one.sweep_many(0, 16, 1)
    assert len(sts) == 16 and all(st.tokens == c.total_tokens and st.aborted_docs == 0 for st in sts)
    assert sts[-1].changed < sts[0].changed                                  # the chain is settling

    # (a) 1-round primary, 16-bit mirror, walk threshold 0.5
    one.set_tuning(force_primary=1, walk_fixed=1, walk_theta=[0.5] * M, narrow=-1)
    o = _prefix_oracle(c, hy, one, inactive)
    _sync_oracle_model(o, one, hy, inactive, M)
    st = one.sweep(16, 1)
    assert st.tokens == c.total_tokens
    o.sweep(16, 1, flags=ORC_NO_APPLY, doc_id_base=0)
    _assert_prefix(o, one, c, f"{name} 1-round / mirror / threshold sweep")
    one.set_tuning(force_primary=0, walk_fixed=0)

    # (b) the segmented sweep: a twin handle takes the same state and runs it as ONE call
    twin = NativeSampler(K, V)
    for m in range(M):
        twin.set_corpus(m, c.doc_off[m], c.tokens[m]); twin.set_assignments(m, one.get_assignments(m))
        twin.set_counts(m, *one.get_counts(m))
    alpha, ina = one.get_alpha()
    hy_now = Hyper(alpha=alpha, alpha_sum=hy.alpha_sum, beta=hy.beta, beta_sum=hy.beta_sum, gamma=hy.gamma, p_a=hy.p_a, p_b=hy.p_b,
                   inactive=ina if inactive is not None else None)
    twin.set_hyper(hy_now)
    nseg = 8
    tot = sum(np.diff(c.doc_off[m]) for m in range(M))
    pos = np.empty(c.D, dtype=np.int64); pos[np.argsort(-tot, kind="stable")] = np.arange(c.D)     # position in the longest-first order
    seg_of_prefix = pos[:PREFIX] % nseg
    tokens_seen = 0
    for sidx in range(nseg):
        _sync_oracle_model(o, one, hy, inactive, M)
        st = one.sweep(17, 1, flags=SWEEP_LIVE_SEGMENTS(nseg) | SWEEP_ONLY_SEGMENT(sidx))
        tokens_seen += st.tokens
        o.sweep_list(17, 1, np.flatnonzero(seg_of_prefix == sidx), flags=ORC_NO_APPLY, doc_id_base=0)
    assert tokens_seen == c.total_tokens
    _assert_prefix(o, one, c, f"{name} segmented sweep, segment by segment")
    st2 = twin.sweep(17, 1, flags=SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(nseg))
    assert st2.tokens == c.total_tokens
    for m in range(M):
        assert np.array_equal(one.get_assignments(m), twin.get_assignments(m)), f"{name}: one SEGMENT_APPLY call differs from 8 single-segment calls in view {m}"
        a, b = one.get_counts(m); a2, b2 = twin.get_counts(m)
        assert np.array_equal(a, a2) and np.array_equal(b, b2)
    assert np.array_equal(one.get_alpha()[0], twin.get_alpha()[0]) and np.array_equal(one.get_alpha()[1], twin.get_alpha()[1])
    twin.close(); o.close()

    # (c) the live sweep
    type_totals = [np.bincount(c.tokens[m], minlength=V[m]).astype(np.int64) for m in range(M)]
    st = one.sweep(18, 1, flags=SWEEP_LIVE)
    assert st.tokens == c.total_tokens and st.new_mass_cnt + st.topic_doc_mass_cnt + st.word_ftree_mass_cnt == st.tokens
    for m in range(M):
        z = one.get_assignments(m)
        nwk, nk = one.get_counts(m)
        assert nwk.min() >= 0 and nk.min() >= 0
        assert np.array_equal(nwk, _recount(c.tokens[m], z, V[m], K)), f"{name}: after a live sweep n_wk is not the count of z in view {m}"
        assert np.array_equal(nwk.astype(np.int64).sum(axis=0), nk.astype(np.int64))
        assert np.array_equal(nwk.astype(np.int64).sum(axis=1), type_totals[m])
    one.close()


def test_c4_driver_command_fingerprint():
    """`python bench.py --steps 20 --warmup 5`: 25 deferred sweeps of C4 from the addInstances start with seed 20260101.  The n_k
    fingerprint of the final counts has been the same in every round's driver record (BENCH_r01.json, BENCH_r02.json) although the
    kernels changed completely in between: a result must not depend on the build, the kernel variants or the walk thresholds."""
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    cfg = synth.CONFIGS["C4"]
    K, V = cfg["K"], cfg["V"]
    c = synth.make_config("C4")
    z0 = init_assignments(K, c.doc_off, seed=1)
    s = NativeSampler(K, V)
    for m in range(3):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(K, V)); s.build_counts()
    s.sweep_many(0, 25, 20260101)
    fp = [int(np.asarray(s.get_counts(m)[1], dtype=np.int64).dot(np.arange(1, K + 1, dtype=np.int64))) for m in range(3)]
    assert fp == [25605983026, 1284819550, 2564199551]
    s.close()
