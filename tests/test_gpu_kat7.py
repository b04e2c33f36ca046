"""KAT-7 on the device: the HIP sweep's per-token conditional against the numpy derivation of tests/kat7.py
(written from SURVEY §8a step 4, no oracle in the expected value), through the C ABI, for both kernels."""
import numpy as np
import pytest

from mvtopicmodel_amd.native import Hyper, SWEEP_GENERIC_KERNEL, SWEEP_EXACT_CHAIN, SWEEP_NO_APPLY
from tests import kat7

pytestmark = pytest.mark.gpu


def _run_native(seed, with_inactive, m_t, pos_t, kflag):
    from mvtopicmodel_amd import NativeSampler
    doc_off, toks, z = kat7.corpus()
    hy = kat7.hyper(with_inactive); p = kat7.view_weights(); nwk, nk = kat7.global_counts()
    s = NativeSampler(kat7.K, kat7.V)
    for m in range(kat7.M):
        s.set_corpus(m, doc_off[m], toks[m]); s.set_assignments(m, z[m])
    s.set_hyper(Hyper(alpha=hy["alpha"], alpha_sum=hy["alpha_sum"], beta=hy["beta"], beta_sum=hy["beta_sum"], gamma=hy["gamma"],
                      p_a=np.full((3, 3), 0.31), p_b=np.ones((3, 3)), inactive=hy["inactive"]))
    for m in range(kat7.M):
        s.set_counts(m, nwk[m], nk[m])
    st = s.sweep(0, seed, flags=kflag | SWEEP_NO_APPLY, p=p, trace=[(0, m_t, pos_t)])
    za = [s.get_assignments(m) for m in range(kat7.M)]
    s.close()
    want, facts = kat7.expected_conditional(hy, p, nwk, nk, kat7.entity_slices(doc_off, z),
                                            kat7.entity_slices(doc_off, za), m_t, pos_t)
    return st.trace[0], want, facts


@pytest.mark.parametrize("kflag", [0, SWEEP_GENERIC_KERNEL, SWEEP_EXACT_CHAIN])
@pytest.mark.parametrize("with_inactive,m_t,pos_t,seed", [(False, 1, 3, 150), (False, 2, 1, 9), (True, 1, 4, 12), (True, 2, 1, 5)])
def test_kat7_device_conditional(with_inactive, m_t, pos_t, seed, kflag):
    got, want, facts = _run_native(seed, with_inactive, m_t, pos_t, kflag)
    assert facts["removed"] and facts["entered_outside_list"] and facts["reentered_removed"]
    assert np.allclose(got, want, rtol=0, atol=1e-12)


def test_kat7_device_every_history():
    worst = 0.0
    for seed in range(12):
        for with_inactive in (False, True):
            for (m_t, pos_t) in ((0, 5), (1, 0), (1, 4), (2, 2)):
                got, want, _ = _run_native(seed, with_inactive, m_t, pos_t, 0)
                worst = max(worst, float(np.abs(got - want).max()))
    assert worst < 1e-12
