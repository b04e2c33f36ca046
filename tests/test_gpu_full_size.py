"""BASELINE config C4 at full size (1 M entities x 3 views, K=400, 147 M tokens) on one GPU: no oracle can follow
at this size, so the sweep is checked through properties that do not depend on it -- conservation of every count
the reference keeps (PTM:511,640-643,872), the sweep statistics, and a checksum of checksums that must not depend on
how the entities are cut into document shards (any sharding gives the same integers)."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper, SWEEP_NO_APPLY

pytestmark = pytest.mark.gpu


def _fingerprint(s, M):
    out = []
    for m in range(M):
        nwk, nk = s.get_counts(m)
        w = (np.arange(nk.size, dtype=np.int64) % 977) + 1
        out.append(int((nk.astype(np.int64) * w).sum()))
        out.append(int((nwk.astype(np.int64).sum(axis=0) * w).sum()))
    return out


def test_c4_full_size_invariants_and_shard_independence():
    import torch
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.dist import GpuShard
    from mvtopicmodel_amd.host import init_assignments
    cfg = synth.CONFIGS["C4"]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config("C4")
    assert c.D == 1_000_000 and 140_000_000 < c.total_tokens < 155_000_000
    z0 = init_assignments(K, c.doc_off, seed=1)                       # PTM:465-515 with java.util.Random(1)
    hy = Hyper.defaults(K, V)
    type_totals = [np.bincount(c.tokens[m], minlength=V[m]).astype(np.int64) for m in range(M)]

    one = NativeSampler(K, V)
    for m in range(M):
        one.set_corpus(m, c.doc_off[m], c.tokens[m]); one.set_assignments(m, z0[m])
    one.set_hyper(hy); one.build_counts()
    for it in range(2):
        st = one.sweep(it, 1)
        assert st.tokens == c.total_tokens and st.aborted_docs == 0 and st.oov_skipped == 0
        assert st.new_mass_cnt + st.topic_doc_mass_cnt + st.word_ftree_mass_cnt == st.tokens     # WRK:33-35
        assert 0.5 * st.tokens < st.changed <= st.tokens
    for m in range(M):
        nwk, nk = one.get_counts(m)
        assert nwk.min() >= 0
        assert np.array_equal(nwk.astype(np.int64).sum(axis=1), type_totals[m])               # PTM:872 typeTotals
        assert np.array_equal(nwk.astype(np.int64).sum(axis=0), nk.astype(np.int64))           # PTM:640-643
        assert int(nk.astype(np.int64).sum()) == int(c.doc_off[m][-1])                          # PTM:511 totalTokens
        z = one.get_assignments(m)
        assert z.min() >= 0 and z.max() < K
        assert np.array_equal(np.bincount(z, minlength=K).astype(np.int64), nk.astype(np.int64))
    want = _fingerprint(one, M)
    z_one = [one.get_assignments(m) for m in range(M)]
    glob0 = None
    one.close()

    # the same two sweeps on two document shards (token-balanced cut, global entity ids), deltas summed on the device
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
    total = gs[0].counts + gs[1].counts                              # all-reduce of the initial counts
    for g in gs:
        g.counts.copy_(total)
    torch.cuda.synchronize()
    for it in range(2):
        for g in gs:
            g.sweep_local(it, 1)
        total = gs[0].delta + gs[1].delta                            # what RCCL's all_reduce(SUM) does across ranks
        for g in gs:
            g.delta.copy_(total)
        torch.cuda.synchronize()
        for g in gs:
            g.apply(-1, -1)
    assert _fingerprint(shards[0], M) == want and _fingerprint(shards[1], M) == want
    for m in range(M):
        zcat = np.concatenate([s.get_assignments(m) for s in shards])
        assert np.array_equal(zcat, z_one[m])
    for s in shards:
        s.close()
