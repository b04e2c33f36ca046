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
    from mvtopicmodel_amd.host import init_assignments
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
