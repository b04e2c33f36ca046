#!/usr/bin/env python3
"""Sweep-kernel time against the walk threshold of the chunk head (SweepLaunch::walk_theta), inside ONE process: the
thresholds change when the trees are walked, never what is sampled, so consecutive sweeps of the same chain can be timed
under different thresholds (MVHDP_WALK_THETA is read by every mvhdp_sweep call).

  python tools/walk_theta_scan.py --workload C4 --thetas 0 0.5 0.6 0.65 0.7 0.75 --rounds 4
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
    ap.add_argument("--burn", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--thetas", nargs="+", default=["0", "0.5", "0.6", "0.65", "0.7", "0.75"],
                    help="each one a threshold for view 0 (other views 0) or a comma list per view, or 'auto'")
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
    it = 0

    def fix(theta):                                  # thresholds through the tuning block (the library reads the environment once, at create)
        if theta == "auto":
            s.set_tuning(walk_fixed=0)
        else:
            th = [float(x) for x in theta.split(",")] if "," in theta else [float(theta)] + [0.0] * (c.M - 1)
            s.set_tuning(walk_fixed=1, walk_theta=th)

    fix("0")
    for _ in range(a.burn):
        s.sweep(it, 1); it += 1
    times = {t: [] for t in a.thetas}
    for r in range(a.rounds):
        order = a.thetas if r % 2 == 0 else a.thetas[::-1]           # forwards and backwards: the chain's drift cancels
        for t in order:
            fix(t)
            st = s.sweep(it, 1); it += 1
            times[t].append(st.sweep_kernel_ms)
    base = sum(times[a.thetas[0]]) / len(times[a.thetas[0]])
    for t in a.thetas:
        v = times[t]
        print(json.dumps({"workload": a.workload, "theta": t, "kernel_ms_mean": round(sum(v) / len(v), 3), "kernel_ms": [round(x, 3) for x in v],
                          "vs_first": round(sum(v) / len(v) / base, 4)}), flush=True)
    s.close()


if __name__ == "__main__":
    main()
