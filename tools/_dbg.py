import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvtopicmodel_amd.native import Hyper, SWEEP_SEGMENT_APPLY, SWEEP_SEGMENT_OVERLAP, SWEEP_LIVE_SEGMENTS
from tests.helpers import make_native, make_oracle, small_corpus
K, V = 300, [2000, 200, 150]
c = small_corpus(K, V, 157, [200, 9, 12], 33)
hy = Hyper.defaults(K, V)
o = make_oracle(c, hy)
z = [o.get_assignments(m) for m in range(c.M)]
a, b = make_native(c, hy, z), make_native(c, hy, z)
sa = a.sweep(0, 5)
sb = b.sweep(0, 5, flags=SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(2))
print('deferred', sa.changed, sa.topic_doc_mass_cnt, 'overlap2', sb.changed, sb.topic_doc_mass_cnt)
tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
order = np.argsort(-tot, kind="stable")
pos = np.empty(c.D, dtype=int); pos[order] = np.arange(c.D)
for m in range(c.M):
    za, zb = a.get_assignments(m), b.get_assignments(m)
    bad = [d for d in range(c.D) if not np.array_equal(za[c.doc_off[m][d]:c.doc_off[m][d+1]], zb[c.doc_off[m][d]:c.doc_off[m][d+1]])]
    print('view', m, 'entities differing', len(bad), [(d, int(pos[d]), int(pos[d]) % 2, int(tot[d])) for d in bad[:12]])
for m in range(c.M):
    (wa, ka), (wb, kb) = a.get_counts(m), b.get_counts(m)
    print('counts equal', np.array_equal(wa, wb), np.array_equal(ka, kb))
