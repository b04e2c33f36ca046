"""BASELINE config C1 (SampleData/SMSSpamCollection2.txt, one view, K = 20): CPU plumbing, SURVEY §8d.  The corpus is
the committed integer fixture tests/golden/c1_smsspam.npz made by tools/c1_smsspam.py from the reference's data file
(token ids, document offsets, vocabulary -- data, no code); where /root/reference is present the loader is re-run
and must reproduce the fixture.  Vocabulary parity with a Java run is not claimed (no reference test pins MALLET's
tokeniser); what is checked is the plumbing: text -> alphabet -> CSR -> addInstances -> sweeps -> counts, LL, top words."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import c1_smsspam  # noqa: E402


def test_fixture_shape_and_loader():
    doc_off, tokens, words = c1_smsspam.load_fixture()
    assert len(doc_off) - 1 == 5574 and doc_off[0] == 0 and doc_off[-1] == len(tokens)
    assert tokens.min() == 0 and tokens.max() == len(words) - 1
    assert len(set(words.tolist())) == len(words)
    # Alphabet order = first seen: a type's first occurrence position is increasing in its id
    first = np.full(len(words), len(tokens), dtype=np.int64)
    np.minimum.at(first, tokens, np.arange(len(tokens)))
    assert np.all(np.diff(first) > 0)
    assert c1_smsspam.simple_tokenize("Free entry in 2 a wkly comp, T&C's apply 0845x", {"in", "a"}) == ["free", "entry", "wkly", "comp", "apply"]
    assert c1_smsspam.simple_tokenize("re_use co-op naïve 4ever", set()) == ["re_use", "naïve", "ever"]      # '_' extends, '-' ends, digits vanish
    if os.path.exists(os.path.join(c1_smsspam.REF, "SampleData", "SMSSpamCollection2.txt")):
        d2, t2, w2, names, labels = c1_smsspam.load()
        assert np.array_equal(d2, doc_off) and np.array_equal(t2, tokens) and w2.tolist() == words.tolist()
        assert set(labels.tolist()) == {"ham", "spam"} and names[0] == "1"


def test_c1_restated_reference_run():
    """K = 20, single view (the LDA path: p[0][0] = 1, no nextBeta call), the reference's thread topology on the CPU."""
    doc_off, tokens, words = c1_smsspam.load_fixture()
    lls, nwk, nk, z, top = c1_smsspam.run_reference_port(doc_off, tokens, words, K=20, iterations=20, threads=4, quiet=True)
    assert lls[-1][1] > lls[0][1] + 0.5 and lls[-1][1] > lls[1][1]                    # PTM:1302-1304 trend
    assert nwk.min() >= 0 and int(nk.sum()) == len(tokens)                               # PTM:511
    assert np.array_equal(nwk.sum(axis=0), nk)                                           # PTM:640-643
    assert np.array_equal(nwk.sum(axis=1), np.bincount(tokens, minlength=len(words)))    # PTM:872
    ref = np.zeros_like(nwk); np.add.at(ref, (tokens, z), 1)
    assert np.array_equal(ref, nwk)
    assert all(len(t) > 0 for t in top)


@pytest.mark.gpu
def test_c1_on_the_gpu_matches_the_oracle():
    """The same corpus through the C ABI: deferred sweeps bit-exact against the oracle, whole corpus."""
    from mvtopicmodel_amd import NativeSampler
    from mvtopicmodel_amd.native import Hyper
    from oracle.binding import Oracle
    doc_off, tokens, words = c1_smsspam.load_fixture()
    K, V = 20, [len(words)]
    hy = Hyper.defaults(K, V)
    o = Oracle(K, V)
    o.set_corpus(0, doc_off, tokens)
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
    o.init_assignments(1); o.build_counts()
    s = NativeSampler(K, V)
    s.set_corpus(0, doc_off, tokens); s.set_assignments(0, o.get_assignments(0))
    s.set_hyper(hy); s.build_counts()
    ll0 = s.model_log_likelihood()[0]
    for it in range(5):
        o.sweep(it, 2024)
        st = s.sweep(it, 2024)
        assert st.tokens == len(tokens)
        assert np.array_equal(o.get_assignments(0), s.get_assignments(0))
        a, b = o.get_counts(0), s.get_counts(0)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert s.model_log_likelihood()[0] > ll0
    assert abs(s.model_log_likelihood()[0] - o.model_log_likelihood()[0]) < 1e-9 * abs(ll0)
    s.close()
