#!/usr/bin/env python3
"""Token-weighted distribution of the entities' topic-list sizes (distinct topics over all views) along a chain: how many tokens sit
in lists of at most 8 / 16 / 32 / 64 / 128 slots after n sweeps.  Decides what a kernel variant for short lists could win.

  python tools/list_size_hist.py --workload C4 --docs 250000 --at 5 10 15 20 25 40
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--docs", type=int, default=250000)
    ap.add_argument("--at", type=int, nargs="+", default=[5, 10, 15, 20, 25, 40])
    a = ap.parse_args()
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper
    c = synth.make_config(a.workload, D=a.docs)
    inactive, K_init = synth.config_inactive(a.workload)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    s = NativeSampler(c.K, c.V)
    for m in range(c.M):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(c.K, c.V, inactive=inactive)); s.build_counts()
    ent = [np.repeat(np.arange(c.D, dtype=np.int64), np.diff(c.doc_off[m])) for m in range(c.M)]
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M)).astype(np.float64)
    it = 0
    for n in sorted(a.at):
        while it < n:
            s.sweep(it, 20260101); it += 1
        keys = np.concatenate([ent[m] * c.K + s.get_assignments(m) for m in range(c.M)])
        keys = np.unique(keys)
        sizes = np.bincount(keys // c.K, minlength=c.D)
        row = {"workload": a.workload, "docs": c.D, "after_sweeps": n, "mean_list": float((sizes * tot).sum() / tot.sum())}
        for b in (8, 16, 32, 64, 128):
            row["tokens_in_lists_le_%d" % b] = round(float(tot[sizes <= b].sum() / tot.sum()), 4)
        print(json.dumps(row), flush=True)
    s.close()


if __name__ == "__main__":
    main()
