"""Diagnostic: per-token latency of the sweep kernels on long entities (one GPU).
   python tools/bench_long_entities.py D LEN K [M]
D entities of LEN text tokens each (plus 8-token side views when M > 1), random types and topics;
prints sweep kernel ms, tokens/s and us per token per entity for a few sweeps."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mvtopicmodel_amd import NativeSampler
from mvtopicmodel_amd.native import Hyper
from mvtopicmodel_amd.synth import Corpus

D, LEN, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1
V = [50000] + [5000] * (M - 1)
rng = np.random.RandomState(1)
lens = [np.full(D, LEN, dtype=np.int64)] + [np.full(D, 8, dtype=np.int64) for _ in range(M - 1)]
offs = [np.concatenate([[0], np.cumsum(l)]) for l in lens]
toks = [rng.randint(0, V[m], offs[m][-1]).astype(np.int32) for m in range(M)]
c = Corpus(K, V, offs, toks)
hy = Hyper.defaults(K, V)
s = NativeSampler(K, V)
for m in range(M):
    s.set_corpus(m, c.doc_off[m], c.tokens[m])
    s.set_assignments(m, rng.randint(0, K, offs[m][-1]).astype(np.int32))
s.set_hyper(hy)
s.build_counts()
tot = int(sum(o[-1] for o in offs))
for it in range(4):
    st = s.sweep(it, 3)
    per_entity_us = st.sweep_kernel_ms * 1e3 / (LEN + 8 * (M - 1))
    print(f"sweep {it}: kernel {st.sweep_kernel_ms:.3f} ms  {tot / st.sweep_kernel_ms / 1e6:.4f} G tok/s  "
          f"{per_entity_us:.3f} us per token of one entity (if entities ran fully in parallel)  fallbacks {st.exact_fallbacks}")
s.close()
