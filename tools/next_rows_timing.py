#!/usr/bin/env python3
"""Wall time of the steps either side of the sweep at BASELINE config C4 on one GPU, through the C ABI:
the statistics kernels behind optimizeBeta / optimizeP / optimizeDP / modelLogLikelihood, the inferencer's device steps and
a frozen sweep, next to the bytes each has to touch (host-side copies of the results included: these calls are synchronous
and return host arrays).  A measurement tool; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_FROZEN
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    cfg = synth.CONFIGS[name]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(name)
    inactive, K_init = synth.config_inactive(name)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    s = NativeSampler(K, V)
    t0 = time.perf_counter()
    for m in range(M):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    upload_s = time.perf_counter() - t0
    s.set_hyper(Hyper.defaults(K, V, inactive=inactive)); s.build_counts()
    for it in range(3):
        s.sweep(it, 1)
    N = c.total_tokens
    counts_bytes = (sum(V) * K + M * K) * 4
    out = {"workload": name, "tokens": N, "entities": c.D,
           "corpus_upload_s": upload_s, "corpus_upload_note": "set_corpus + set_assignments of every view from pageable host arrays "
           "(the only time corpus data crosses PCIe): %.2f GB" % (N * 8 / 1e9)}

    def timed(label, fn, bytes_touched, reps=3):
        fn()
        t = []
        for _ in range(reps):
            a = time.perf_counter(); fn(); t.append(time.perf_counter() - a)
        ms = min(t) * 1e3
        out[label] = {"ms": round(ms, 3), "bytes_touched_MB": round(bytes_touched / 1e6, 1), "GB_per_s": round(bytes_touched / ms / 1e6, 1)}

    timed("modelLogLikelihood (PTM:3322-3452), all views", s.model_log_likelihood, N * 4 + counts_bytes)
    maxc = 1 << 16
    timed("countHistogram of optimizeBeta (PTM:2295-2309), view 0", lambda: s.get_count_histogram(0, maxc), V[0] * K * 4)
    if M > 1:
        timed("pDistr_Mean sums of optimizeP (PTM:2706-2792)", s.view_overlap_sums, N * 4 + c.D * M * M * 8)
    hl = int(max(np.diff(c.doc_off[0]).max(), 1)) + 1
    timed("topicDocCounts / docLengthCounts of optimizeDP (PTM:620-651), view 0", lambda: s.get_doc_topic_hist(0, hl, hl), int(c.doc_off[0][-1]) * 4 + K * hl * 4)
    w = np.ones(M)
    nd = min(c.D, 100000)
    timed("printDocumentTopics proportions (PTM:2871-2899), %d entities" % nd, lambda: s.doc_topic_proportions(w, 0, nd), nd * K * 8 + N * 4 * nd // c.D)
    timed("buildFTrees (PTM:2660-2696), every tree + descent table", s.build_trees, counts_bytes + sum(V) * 2 * K * 8 * 2)
    timed("initInferencer trees (INF:557-586)", s.build_inference_trees, counts_bytes + sum(V) * 2 * K * 8 * 2)
    timed("inferencer: topics drawn from the trees (INF:169-199)", lambda: s.init_assignments_from_trees(7), N * 8)
    st = {}
    def frozen():
        st["s"] = s.sweep(99, 7, flags=SWEEP_FROZEN)
    timed("frozen sweep (INF:211-294: nst = 1, nut = 0)", frozen, N * (4 * K + 8))
    out["frozen sweep tokens/s"] = round(st["s"].tokens / (out["frozen sweep (INF:211-294: nst = 1, nut = 0)"]["ms"] / 1e3) / 1e9, 3)
    print(json.dumps(out, indent=1))
    s.close()


if __name__ == "__main__":
    main()
