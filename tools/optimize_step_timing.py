import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from mvtopicmodel_amd import synth
from hostmirror.binding import FastQMVWVParallelTopicModel
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[name]; K, V = cfg["K"], cfg["V"]; M = len(V)
c = synth.make_config(name)
training = []
for m in range(M):
    lens = np.diff(c.doc_off[m]); have = np.flatnonzero(lens > 0)
    off = np.concatenate([[0], np.cumsum(lens[have])]).astype(np.int64)
    training.append((have.astype(np.int64), off, c.tokens[m], V[m]))
model = FastQMVWVParallelTopicModel(K, M, 0.1, 0.01)
model.setNumIterations(30); model.setBurninPeriod(100); model.setOptimizeInterval(50); model.setRandomSeed(1)
if len(sys.argv) > 2 and sys.argv[2] == "device":
    model.setDeviceGammaStatistics(True); model.setDeviceTableStatistics(True)
model.addInstances(training); model.estimate()
out = {}
for nm, fn in (("optimizeP", model.optimizeP), ("optimizeDP", model.optimizeDP), ("optimizeGamma", model.optimizeGamma), ("optimizeBeta", model.optimizeBeta), ("modelLogLikelihood", model.modelLogLikelihood)):
    t0 = time.perf_counter(); fn(); out[nm] = round((time.perf_counter() - t0) * 1e3, 1)
print(name, json.dumps(out))
model.close()
