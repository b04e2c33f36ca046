#!/usr/bin/env python3
"""Kernel / total ms per sweep of one chain in the three update modes, for A/B runs of two library builds on one box.

  python tools/mode_times.py --workload C4 --mode seg8 --sweeps 40        (mode: deferred | live4 | live1 | seg8 | seg4 | oseg8: overlapped segments)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--mode", default="deferred")
    ap.add_argument("--sweeps", type=int, default=40)
    a = ap.parse_args()
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_SEGMENT_APPLY, SWEEP_SEGMENT_OVERLAP
    flags = 0
    reuse = False
    if a.mode.startswith("liveR"):
        from mvtopicmodel_amd.native import SWEEP_REUSE_TREES
        flags = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(int(a.mode[5:])) | SWEEP_REUSE_TREES
        reuse = True
    elif a.mode.startswith("live"):
        flags = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(int(a.mode[4:]))
    elif a.mode.startswith("seg"):
        flags = SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(int(a.mode[3:]))
    elif a.mode.startswith("oseg"):
        flags = SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(int(a.mode[4:]))
    c = synth.make_config(a.workload)
    inactive, K_init = synth.config_inactive(a.workload)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    s = NativeSampler(c.K, c.V)
    for m in range(c.M):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(c.K, c.V, inactive=inactive)); s.build_counts()
    ks, ts = [], []
    if a.mode == "atomic_probe":
        # a -DMVHDP_PROBE build of the library: sweeps whose chunk-end atomics go to a cache-resident table (1), drop the +1 (2), are plain
        # stores (3) or workgroup-scope atomics (4), between deferred sweeps of a chain at two ages; the counts are restored after each probe
        from mvtopicmodel_amd.native import SWEEP_NO_APPLY
        out = {"workload": a.workload, "mode": a.mode}
        for start in (6, a.sweeps):
            while len(ks) < start:
                st = s.sweep(len(ks), 20260101); ks.append(round(st.sweep_kernel_ms, 3))
            z = [s.get_assignments(m) for m in range(c.M)]
            res = {"deferred_before": ks[start - 3:start]}
            for kind, name in ((0, "normal"), (1, "near_table"), (2, "minus_only"), (3, "plain_stores"), (4, "workgroup_scope"), (7, "table_4MB"), (5, "table_32MB"), (6, "table_64MB"), (0, "normal_again")):
                for m in range(c.M):
                    s.set_assignments(m, z[m])
                s.build_counts()
                os.environ["MVHDP_ATOMIC_PROBE"] = str(kind)
                st = s.sweep(len(ks), 20260101, flags=SWEEP_NO_APPLY)          # (the probes' deltas are wrong by design: never applied)
                res[name] = round(st.sweep_kernel_ms, 3)
            os.environ["MVHDP_ATOMIC_PROBE"] = "0"
            for m in range(c.M):
                s.set_assignments(m, z[m])
            import time
            t0 = time.perf_counter(); s.build_counts(); res["build_counts_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)   # one far atomic per token
            out[f"after_{start}_sweeps"] = res
        print(json.dumps(out)); s.close(); return
    if a.mode == "frozen_probe":
        # what the chunk-end atomics cost: a settled chain, then frozen sweeps (MVHDP_SWEEP_FROZEN: same sampling, no deltas) between deferred ones
        from mvtopicmodel_amd.native import SWEEP_FROZEN, SWEEP_REUSE_TREES
        out = {"workload": a.workload, "mode": a.mode}
        for start in (6, a.sweeps):
            it0 = len(ks)
            while len(ks) < start:
                st = s.sweep(len(ks), 20260101); ks.append(round(st.sweep_kernel_ms, 3))
            z = [s.get_assignments(m) for m in range(c.M)]
            fr = []
            s.build_trees()                              # (a frozen sweep samples against stored trees: those of the current counts, so the mirror is current too)
            for j in range(4):
                st = s.sweep(1000 + j, 20260101, flags=SWEEP_FROZEN); fr.append(round(st.sweep_kernel_ms, 3))
            for m in range(c.M):
                s.set_assignments(m, z[m])
            s.build_counts()                             # (the same counts: the frozen sweeps changed z only)
            nx = []
            for j in range(3):
                st = s.sweep(len(ks), 20260101); ks.append(round(st.sweep_kernel_ms, 3)); nx.append(ks[-1])
            out[f"after_{start}_sweeps"] = {"deferred_before": ks[start - 3:start], "frozen": fr, "deferred_after": nx}
        print(json.dumps(out)); s.close(); return
    for it in range(a.sweeps):
        extra = 0.0
        if reuse:                                   # the trees of the whole sweep, built by the host's call (timed with the sweep)
            import time
            t0 = time.perf_counter(); s.build_trees(); extra = (time.perf_counter() - t0) * 1e3
        st = s.sweep(it, 20260101, flags=flags)
        ks.append(round(st.sweep_kernel_ms, 3)); ts.append(round(st.total_ms + extra, 3))
    h = len(ts) // 2
    print(json.dumps({"workload": a.workload, "mode": a.mode, "total_ms_sweeps_5_24": round(sum(ts[5:25]) / 20, 3),
                      "total_ms_last_half": round(sum(ts[h:]) / (len(ts) - h), 3), "total_ms": ts, "kernel_ms": ks}))
    s.close()


if __name__ == "__main__":
    main()
