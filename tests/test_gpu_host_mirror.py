"""The C++ host-side mirror of FastQMVWVParallelTopicModel (csrc/host/), driven the way a MALLET
client drives the reference: new ...(K, M, alpha, beta); set*; addInstances(InstanceList[]); estimate()."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper

pytestmark = pytest.mark.gpu


def _instance_lists(rng, V):
    """Three views whose instances are NOT aligned: names decide the entity (PTM:437-455)."""
    def mk(names, lens, v):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return (np.array(names, dtype=np.int64), off, rng.randint(0, v, off[-1]).astype(np.int32), v)
    v0 = mk([10, 11, 12, 13, 14, 15], [30, 1, 45, 9, 0, 22], V[0])
    v1 = mk([12, 99, 10, 15], [4, 6, 3, 5], V[1])          # 99 has no text view -> appended as a new entity
    v2 = mk([99, 13, 77], [2, 8, 1], V[2])                   # 77 appended too
    return [v0, v1, v2]


def test_add_instances_and_estimate_match_the_oracle():
    from mvtopicmodel_amd.host import FastQMVWVParallelTopicModel
    from oracle.binding import Oracle
    K, V = 25, [300, 40, 30]
    rng = np.random.RandomState(3)
    training = _instance_lists(rng, V)
    model = FastQMVWVParallelTopicModel(K, 3, 0.1, 0.01)
    model.setNumIterations(4); model.setBurninPeriod(200); model.setOptimizeInterval(50); model.setRandomSeed(7)
    model.addInstances(training)

    # entity order: view-0 instances in list order, then unmatched instances of later views (PTM:443-455)
    ids0, off0, tok0, z0 = model.get_view(0)
    assert ids0.tolist() == [10, 11, 12, 13, 14, 15, 99, 77]
    assert np.diff(off0).tolist() == [30, 1, 45, 9, 0, 22, 0, 0]
    ids1, off1, tok1, z1 = model.get_view(1)
    assert np.diff(off1).tolist() == [3, 0, 4, 0, 0, 5, 6, 0]
    ids2, off2, tok2, z2 = model.get_view(2)
    assert np.diff(off2).tolist() == [0, 0, 0, 8, 0, 0, 2, 1]
    # tokens landed with their entity
    assert np.array_equal(tok1[off1[2]:off1[3]], training[1][2][0:4])      # name 12 is instance 0 of view 1

    # oracle on the same flattened corpus: same initial draw (java.util.Random(7)) and same counts
    o = Oracle(K, V)
    views = [(off0, tok0), (off1, tok1), (off2, tok2)]
    for m, (off, tok) in enumerate(views):
        o.set_corpus(m, off, tok)
    hy = Hyper.defaults(K, V, p_a=0.2)                       # PTM:1055-1058
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
    o.init_assignments(7)
    for m, z in enumerate((z0, z1, z2)):
        assert np.array_equal(o.get_assignments(m), z), f"initial assignments differ in view {m}"
    o.build_counts()
    for m in range(3):
        a, b = model.get_counts(m)
        assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])

    model.estimate()
    for it in range(1, 5):                                   # PTM:1146,1166-1171
        hy.p_a[:] = min(it / 100 + 0.3, 1.1)
        o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
        o.sweep(it, 7)
    for m in range(3):
        assert np.array_equal(model.get_view(m)[3], o.get_assignments(m)), f"assignments differ in view {m}"
        a, b = model.get_counts(m)
        assert np.array_equal(a, o.get_counts(m)[0]) and np.array_equal(b, o.get_counts(m)[1])
    log = model.iteration_log()
    assert len(log) == 4
    total = int(off0[-1] + off1[-1] + off2[-1])
    for ms, st in log:
        assert st["tokens"] == total
        assert st["new_mass_cnt"] + st["topic_doc_mass_cnt"] + st["word_ftree_mass_cnt"] == total
    model.close()


def test_single_view_is_plain_lda_path():
    """BASELINE config 1/2 shape: M=1 (no view weights are drawn, p[0][0]=1, KAT-6)."""
    from mvtopicmodel_amd.host import FastQMVWVParallelTopicModel, init_assignments
    K, V = 20, [500]
    rng = np.random.RandomState(4)
    lens = rng.poisson(20, 80) + 1
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tok = rng.randint(0, 500, off[-1]).astype(np.int32)
    model = FastQMVWVParallelTopicModel(K, 1, 0.1, 0.01)
    model.setNumIterations(3); model.setRandomSeed(11)
    model.addInstances([(np.arange(80, dtype=np.int64), off, tok, 500)])
    z_init = model.get_view(0)[3].copy()
    assert np.array_equal(z_init, init_assignments(K, [off], 11)[0])
    model.estimate()
    nwk, nk = model.get_counts(0)
    z = model.get_view(0)[3]
    ref = np.zeros_like(nwk); np.add.at(ref, (tok, z), 1)
    assert np.array_equal(ref, nwk) and np.array_equal(nwk.sum(axis=0), nk)
    assert (z != z_init).mean() > 0.3
    model.close()
