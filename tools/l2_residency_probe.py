import sys
sys.path.insert(0, '.')
import numpy as np
from mvtopicmodel_amd import NativeSampler, synth
from mvtopicmodel_amd.java_init import init_assignments
from mvtopicmodel_amd.native import Hyper
for V in ([50000, 5000, 5000], [5000, 500, 500], [500, 50, 50]):
    K, D = 400, 300000
    c = synth.generate(K, V, D, [127, 7, 15], 0x5EED0004, name="x")
    z0 = init_assignments(K, c.doc_off, seed=1)
    s = NativeSampler(K, V)
    for m in range(3):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(K, V)); s.build_counts()
    for it in range(5): s.sweep(it, 1)
    ms = []
    for it in range(5, 15):
        st = s.sweep(it, 1); ms.append(st.sweep_kernel_ms)
    print(V, "tokens", c.total_tokens, "kernel ms", round(float(np.mean(ms)), 3), "G tok/s", round(c.total_tokens / np.mean(ms) / 1e6, 3), "n_wk MB", sum(V) * K * 4 / 1e6, flush=True)
    s.close()
