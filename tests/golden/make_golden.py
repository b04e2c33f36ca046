#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/).  The reference itself cannot be
run (Java, no JVM in the image) and ships no vectors, so these fixtures pin the oracle and the
HIP path against regressions of each other; they are not reference-produced outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mvtopicmodel_amd import synth                      # noqa: E402
from mvtopicmodel_amd.native import Hyper               # noqa: E402
from oracle.binding import Oracle                       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: K, V, D, lam, corpus seed, sweep seed, inactive topics
    "m1_k5": (5, [37], 64, [11], 101, 7, None),
    "m1_k100": (100, [900], 64, [90], 102, 8, None),
    "m3_k20": (20, [250, 40, 30], 64, [30, 4, 6], 103, 9, None),
    "m3_k100_inactive": (100, [800, 90, 70], 64, [60, 7, 9], 104, 10, [97, 98, 99]),
    # C5 shape: five views, K=1000, power-law text lengths (a few entities with several hundred topics: the wide kernel
    # variants and both dispatch modes see this one)
    "m5_k1000_powerlaw": (1000, [4000, 300, 300, 300, 300], 96, [96, 7, 7, 7, 7], 105, 11, None),
}
POWER_LAW = {"m5_k1000_powerlaw"}


def build(name):
    K, V, D, lam, cseed, sseed, inactive = CASES[name]
    c = synth.generate(K, V, D, lam, cseed, chunk_docs=4096, power_law_text=name in POWER_LAW)
    ina = None
    if inactive:
        ina = np.zeros(K, dtype=np.uint8); ina[inactive] = 1
    hy = Hyper.defaults(K, V, inactive=ina)
    if inactive:
        hy.alpha[:, K] = 12.0
    o = Oracle(K, V)
    for m in range(c.M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    o.init_assignments(1)
    z0 = [o.get_assignments(m) for m in range(c.M)]
    if inactive:
        for m in range(c.M):
            z0[m][np.isin(z0[m], inactive)] = 0
            o.set_assignments(m, z0[m])
    o.build_counts()
    trace = [(d, m, 0) for d in (0, 17, 63) for m in range(c.M) if c.doc_off[m][d + 1] > c.doc_off[m][d]]
    out = dict(K=K, V=np.array(V), sweep_seed=sseed, trace=np.array(trace, dtype=np.int64),
               alpha=hy.alpha, alpha_sum=hy.alpha_sum, beta=hy.beta, beta_sum=hy.beta_sum, gamma=hy.gamma,
               p_a=hy.p_a, p_b=hy.p_b, inactive=(ina if ina is not None else np.zeros(K, dtype=np.uint8)))
    for m in range(c.M):
        out[f"doc_off{m}"] = c.doc_off[m]; out[f"tokens{m}"] = c.tokens[m]; out[f"z0_{m}"] = z0[m]
    for it in range(3):
        r = o.sweep(it, sseed, want_dbg=(it == 0), trace=trace if it == 0 else None)
        st = r["stats"]
        out[f"stats{it}"] = np.array([st[k] for k in ("tokens", "changed", "new_mass_cnt", "topic_doc_mass_cnt",
                                                      "word_ftree_mass_cnt", "activated_topic", "activated_modality")], dtype=np.int64)
        if it == 0:
            out["trace_probs"] = r["trace"]
            for m in range(c.M):
                out[f"dbg{m}"] = r["dbg"][m]
        for m in range(c.M):
            out[f"z{it + 1}_{m}"] = o.get_assignments(m)
            nwk, nk = o.get_counts(m)
            out[f"nk{it + 1}_{m}"] = nk
            if it in (0, 2):
                out[f"nwk{it + 1}_{m}"] = nwk
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return out


if __name__ == "__main__":
    for name in CASES:
        build(name)
        print("wrote", name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
