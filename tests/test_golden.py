"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the
oracle): the oracle must keep reproducing them (CPU) and the HIP path must reproduce them (GPU)."""
import glob
import os

import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz"))
               if not os.path.basename(f).startswith("c1_"))      # c1_*.npz is a corpus (tests/test_c1_plumbing.py), not a sweep vector
STAT_KEYS = ("tokens", "changed", "new_mass_cnt", "topic_doc_mass_cnt", "word_ftree_mass_cnt",
             "activated_topic", "activated_modality")


def _load(path):
    g = np.load(path)
    K = int(g["K"]); V = [int(v) for v in g["V"]]
    hy = Hyper(alpha=g["alpha"], alpha_sum=g["alpha_sum"], beta=g["beta"], beta_sum=g["beta_sum"],
               gamma=g["gamma"], p_a=g["p_a"], p_b=g["p_b"], inactive=g["inactive"])
    return g, K, V, hy


def _check(g, M, it, stats, get_z, get_counts):
    assert [int(stats[k]) for k in STAT_KEYS] == g[f"stats{it}"].tolist()
    for m in range(M):
        assert np.array_equal(get_z(m), g[f"z{it + 1}_{m}"])
        nwk, nk = get_counts(m)
        assert np.array_equal(nk, g[f"nk{it + 1}_{m}"])
        if it in (0, 2):
            assert np.array_equal(nwk, g[f"nwk{it + 1}_{m}"])


def test_fixtures_exist():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path):
    from oracle.binding import Oracle
    g, K, V, hy = _load(path)
    M = len(V)
    o = Oracle(K, V)
    for m in range(M):
        o.set_corpus(m, g[f"doc_off{m}"], g[f"tokens{m}"])
        o.set_assignments(m, g[f"z0_{m}"])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    o.build_counts()
    trace = [tuple(int(x) for x in t) for t in g["trace"]]
    for it in range(3):
        r = o.sweep(it, int(g["sweep_seed"]), want_dbg=(it == 0), trace=trace if it == 0 else None)
        _check(g, M, it, r["stats"], o.get_assignments, o.get_counts)
        if it == 0:
            assert np.array_equal(r["trace"], g["trace_probs"])
            for m in range(M):
                assert np.array_equal(r["dbg"][m], g[f"dbg{m}"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_reproduces_golden(path):
    from mvtopicmodel_amd import NativeSampler
    from mvtopicmodel_amd.native import SWEEP_EXACT_CHAIN
    g, K, V, hy = _load(path)
    M = len(V)
    trace = [tuple(int(x) for x in t) for t in g["trace"]]
    for flags in (0, SWEEP_EXACT_CHAIN):
        s = NativeSampler(K, V)
        for m in range(M):
            s.set_corpus(m, g[f"doc_off{m}"], g[f"tokens{m}"])
            s.set_assignments(m, g[f"z0_{m}"])
        s.set_hyper(hy)
        s.build_counts()
        for it in range(3):
            r = s.sweep(it, int(g["sweep_seed"]), flags=flags, want_dbg=(it == 0), trace=trace if it == 0 else None)
            stats = {k: getattr(r, k) for k in STAT_KEYS}
            _check(g, M, it, stats, s.get_assignments, s.get_counts)
            if it == 0:
                # north_star tolerance: per-token conditional probabilities within 1e-6
                assert np.max(np.abs(r.trace - g["trace_probs"])) < 1e-6
                for m in range(M):
                    if flags & SWEEP_EXACT_CHAIN:
                        assert np.array_equal(r.dbg[m], g[f"dbg{m}"])      # sequential sum: bit-identical masses
                    else:
                        assert np.allclose(r.dbg[m], g[f"dbg{m}"], rtol=1e-12, atol=0)
        s.close()
