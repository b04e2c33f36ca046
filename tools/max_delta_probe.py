import sys; sys.path.insert(0,'.')
import numpy as np, torch
from mvtopicmodel_amd import NativeSampler, synth
from mvtopicmodel_amd.dist import GpuShard
from mvtopicmodel_amd.java_init import init_assignments
from mvtopicmodel_amd.native import Hyper
N=int(sys.argv[1]); name=sys.argv[2] if len(sys.argv)>2 else "C4"
cfg=synth.CONFIGS[name]; K,V=cfg["K"],cfg["V"]; M=len(V)
c=synth.make_config(name); ina,Ki=synth.config_inactive(name); z0=init_assignments(Ki,c.doc_off,seed=1)
tot=sum(np.diff(c.doc_off[m]) for m in range(M)); bounds=synth.shard_bounds(tot,N)
sh=[]
for lo,hi in bounds:
    sub=c.slice_docs(lo,hi); s=NativeSampler(K,V,doc_id_base=lo)
    for m in range(M): s.set_corpus(m,sub.doc_off[m],sub.tokens[m]); s.set_assignments(m,z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
    s.set_hyper(Hyper.defaults(K,V,inactive=ina)); s.build_counts(); sh.append(GpuShard(s,"cuda:0"))
torch.cuda.synchronize(); total=sh[0].counts.clone()
for g in sh[1:]: total+=g.counts
for g in sh: g.counts.copy_(total)
torch.cuda.synchronize()
for g in sh: g.counts_written()
for it in range(30):
    mx=[]
    for g in sh:
        g.sweep_local(it,1); torch.cuda.synchronize(); mx.append(int(g.delta.abs().max().item()))
    total=sh[0].delta.clone()
    for g in sh[1:]: total+=g.delta
    gm=int(total.abs().max().item())
    for g in sh: g.delta.copy_(total)
    torch.cuda.synchronize()
    for g in sh: g.apply(-1,-1)
    print(it, "max |delta| per shard", max(mx), "of the sum", gm, flush=True)
