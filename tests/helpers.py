"""Shared test helpers: build the same small model in the oracle and in the product."""
import numpy as np

from mvtopicmodel_amd.native import Hyper
from mvtopicmodel_amd import synth


def small_corpus(K, V, D, lam, seed, **kw):
    return synth.generate(K, V, D, lam, seed, chunk_docs=4096, **kw)


def make_oracle(corpus, hyper, init_seed=1):
    from oracle.binding import Oracle
    o = Oracle(corpus.K, corpus.V)
    for m in range(corpus.M):
        o.set_corpus(m, corpus.doc_off[m], corpus.tokens[m])
    o.set_hyper(hyper.alpha, hyper.alpha_sum, hyper.beta, hyper.beta_sum, hyper.gamma,
                hyper.p_a, hyper.p_b, hyper.inactive)
    o.init_assignments(init_seed)
    o.build_counts()
    return o


def make_native(corpus, hyper, z_init, doc_id_base=0):
    from mvtopicmodel_amd import NativeSampler
    s = NativeSampler(corpus.K, corpus.V, device=0, doc_id_base=doc_id_base)
    for m in range(corpus.M):
        s.set_corpus(m, corpus.doc_off[m], corpus.tokens[m])
        s.set_assignments(m, z_init[m])
    s.set_hyper(hyper)
    s.build_counts()
    return s


def assert_same_state(o, s, M):
    for m in range(M):
        zo, zs = o.get_assignments(m), s.get_assignments(m)
        assert np.array_equal(zo, zs), f"z differs in view {m}: {np.count_nonzero(zo != zs)} of {len(zo)}"
        nwk_o, nk_o = o.get_counts(m)
        nwk_s, nk_s = s.get_counts(m)
        assert np.array_equal(nk_o, nk_s), f"n_k differs in view {m}"
        assert np.array_equal(nwk_o, nwk_s), f"n_wk differs in view {m}"
