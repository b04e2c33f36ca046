#!/usr/bin/env python3
"""How predictable is the tree branch (WRK:533-535) at the head of a token chunk?

The sweep kernel walks every token's word tree speculatively (three 128-byte lines per token) although only ~22 % of the
C4 tokens take that branch.  A token takes it iff u1*(newMass + mass + root) - newMass >= mass; u1 and root are known
before the entity's state is touched, mass is not.  This probe dumps (newMass, mass, root, s0) of every token of one
sweep (debug output of mvhdp_sweep) and tabulates, for the rule "walk the tree up front iff u1 >= theta", how many walks
it saves and how many tokens would have to walk on demand inside the serial loop.

  python tools/tree_branch_probe.py --workload C4 --docs 100000 --burn 15
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
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--burn", type=int, default=15)
    ap.add_argument("--out", default=None)
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
    for it in range(a.burn):
        s.sweep(it, 1)
    st = s.sweep(a.burn, 1, want_dbg=True)
    res = {"workload": a.workload, "docs": c.D, "burn": a.burn, "views": []}
    for m in range(c.M):
        d = st.dbg[m]
        new, mass, root, s0 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
        total = new + mass + root
        ok = total > 0
        u1 = s0[ok] / total[ok]
        tree = (s0[ok] - new[ok]) >= mass[ok]
        rows = []
        for th in (0.0, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
            skip = u1 < th
            rows.append({"theta": th, "walks_saved": float(skip.mean()), "on_demand": float((skip & tree).mean())})
        # the bound that needs no tuning: mass >= (share of the entity's mass that does not depend on the word)?  report
        # the quantiles of rho = mass / root instead, which is what any predictor has to guess
        rho = mass[ok] / np.maximum(root[ok], 1e-300)
        res["views"].append({"view": m, "tokens": int(ok.sum()), "tree_frac": float(tree.mean()), "rule_u1_ge_theta": rows,
                             "rho_quantiles": {str(q): float(np.quantile(rho, q)) for q in (0.01, 0.05, 0.1, 0.25, 0.5, 0.75, 0.9)}})
        print(json.dumps(res["views"][-1]))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    s.close()


if __name__ == "__main__":
    main()
