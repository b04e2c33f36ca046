#!/usr/bin/env python3
"""Sweep-kernel and total ms of every sweep of a chain (with MVHDP_DEBUG=1 the library adds its per-sweep choices on stderr).

  python tools/per_sweep_times.py --workload C4 --sweeps 40
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--docs", type=int, default=None)
    ap.add_argument("--sweeps", type=int, default=40)
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
    ks, ts = [], []
    for it in range(a.sweeps):
        st = s.sweep(it, 20260101)
        ks.append(round(st.sweep_kernel_ms, 3)); ts.append(round(st.total_ms, 3))
    print(json.dumps({"workload": a.workload, "walk": os.environ.get("MVHDP_WALK_THETA", "auto"), "kernel_ms": ks, "total_ms": ts,
                      "mean_kernel_ms_last_half": round(sum(ks[len(ks) // 2:]) / (len(ks) - len(ks) // 2), 3)}))
    s.close()


if __name__ == "__main__":
    main()
