import sys; sys.path.insert(0,'.')
import numpy as np
from mvtopicmodel_amd import NativeSampler, synth
from mvtopicmodel_amd.host import init_assignments
from mvtopicmodel_amd.native import Hyper
cfg=synth.CONFIGS["C4"]; K,V=cfg["K"],cfg["V"]; M=3
c=synth.make_config("C4"); z0=init_assignments(K,c.doc_off,seed=1)
s=NativeSampler(K,V)
for m in range(M): s.set_corpus(m,c.doc_off[m],c.tokens[m]); s.set_assignments(m,z0[m])
s.set_hyper(Hyper.defaults(K,V)); s.build_counts()
for it in range(40):
    st=s.sweep(it,20260101)
    print(it, round(st.sweep_kernel_ms,2), round(st.total_ms,2), flush=True)
