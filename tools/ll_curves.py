#!/usr/bin/env python3
"""Is a deferred (snapshot) sweep worth a reference sweep?  LL/token per view against the number of sweeps for

  cpu       the reference's update discipline restated on the CPU (oracle/ref_threaded.c: sampler threads read the
            counts and trees while updater threads apply the deltas, PTM:1036-1101, UPD:164-297)
  deferred  the GPU sweep under the parity contract (every token sampled against the sweep-start counts)
  live      the GPU sweep with MVHDP_SWEEP_LIVE (atomics on the shared counts; the tree branch from the live count rows, or -- live_rows 0 --
            from stored trees rebuilt n times per sweep)

all from the same corpus, the same initial assignments (PTM:465-515 with java.util.Random(1)) and the same fixed
hyper-parameters (alpha 0.1, beta 0.01, gamma 1, p_a 0.31 = iteration 1 of the burn-in schedule, PTM:1168).  The
log-likelihood is modelLogLikelihood (PTM:3322-3452) in both implementations (they agree to 1e-12, tests/test_next_rows.py).

  python tools/ll_curves.py cpu --workload C3 --docs 200000 --sweeps 100 --every 5 --out profiles/r02_ll_cpu.json
  python tools/ll_curves.py gpu --workload C3 --docs 200000 --sweeps 100 --every 5 --out gpurun_out/r02_ll_gpu.json
  python tools/ll_curves.py table profiles/r02_ll_cpu.json profiles/r02_ll_gpu.json > profiles/r02_ll_curves.md

This is a measurement tool (it uses the oracle for the CPU leg), not product code.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(workload, docs):
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.native import Hyper
    c = synth.make_config(workload, D=docs)
    inactive, K_init = synth.config_inactive(workload)
    hy = Hyper.defaults(c.K, c.V, inactive=inactive)
    return c, hy, K_init


def run_cpu(args):
    from oracle.binding import Oracle
    c, hy, K_init = load(args.workload, args.docs)
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, hy.inactive)
    if K_init == c.K:
        o.init_assignments(1)
    else:
        from mvtopicmodel_amd.java_init import init_assignments
        z0 = init_assignments(K_init, c.doc_off, seed=1)
        for m in range(c.M):
            o.set_assignments(m, z0[m])
    o.build_counts()
    ntok = np.array([int(c.doc_off[m][-1]) for m in range(c.M)], dtype=np.float64)
    T = args.threads
    curve = [{"sweep": 0, "ll_per_token": (o.model_log_likelihood() / ntok).tolist()}]
    secs = 0.0
    for it in range(1, args.sweeps + 1):
        # one iteration per call: every call starts at iteration 1 of the burn-in schedule (p_a = 0.31) and rebuilds
        # the trees first, like the GPU sweeps it is compared with
        s, st = o.threaded_estimate(T, 1, args.seed + it)
        secs += s
        if it % args.every == 0 or it == args.sweeps:
            curve.append({"sweep": it, "ll_per_token": (o.model_log_likelihood() / ntok).tolist(),
                          "changed_frac": st["changed"] / max(1, st["tokens"])})
            print(f"cpu sweep {it}: LL/token {curve[-1]['ll_per_token']}  ({secs:.0f} s)", flush=True)
    out = {"runs": {f"cpu reference topology ({3 * T // 4} samplers + {T // 4} updaters)" + (f" seed {args.seed}" if args.seed != 20260101 else ""): curve},
           "workload": args.workload, "docs": c.D, "tokens": c.total_tokens, "cpu_seconds": secs}
    json.dump(out, open(args.out, "w"), indent=1)


def run_gpu(args):
    from mvtopicmodel_amd import NativeSampler
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_REUSE_TREES, SWEEP_SEGMENT_APPLY, SWEEP_SEGMENT_OVERLAP
    c, hy, K_init = load(args.workload, args.docs)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    ntok = np.array([int(c.doc_off[m][-1]) for m in range(c.M)], dtype=np.float64)
    runs = {}
    modes = [("gpu deferred (snapshot sweep)", 0)]
    for n in args.live_segments:
        # (n = 0: the library's default -- one segment in the live-rows form, four with stored trees)
        form = {None: "", 1: " (tree branch from the live rows)", 0: " (stored trees rebuilt per segment)"}[args.live_rows]
        modes.append((f"gpu live, {n if n else 'default'} segment(s) per sweep{form}", SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(n)))
    for n in args.segmented:
        modes.append((f"gpu deferred in {n} segments, applied in between", SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(n)))
    for n in args.overlapped:
        modes.append((f"gpu deferred in {n} overlapped segments (counts two segments behind, trees of the sweep start)",
                      SWEEP_SEGMENT_APPLY | SWEEP_SEGMENT_OVERLAP | SWEEP_LIVE_SEGMENTS(n)))
    for n in args.live_trees_once:
        modes.append((f"gpu live, {n} segments, trees once per sweep", SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(n) | SWEEP_REUSE_TREES))
    if args.only:
        modes = [mo for mo in modes if any(o in mo[0] for o in args.only)]
    # every mode, for every pinned live16 setting and every seed asked for (several seeds of one mode = the noise band of a chain)
    l16s = args.live16 if args.live16 else [None]
    seeds = args.seeds if args.seeds else [args.seed]
    for name0, flags in modes:
      for l16 in l16s:
        if l16 is not None and not (flags & SWEEP_LIVE):
            if l16 != l16s[0]:
                continue                       # (live16 only concerns live sweeps: one run of the others)
        for seed in seeds:
            s = NativeSampler(c.K, c.V)
            for m in range(c.M):
                s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
            s.set_hyper(hy); s.build_counts()
            name = name0
            if l16 is not None and (flags & SWEEP_LIVE):
                s.set_tuning(live16=l16)
                name += " (live16=%d)" % l16
            if args.live_rows is not None and (flags & SWEEP_LIVE):
                s.set_tuning(live_rows=args.live_rows)
            if len(seeds) > 1:
                name += " seed %d" % seed
            if args.tag:
                name += " [%s]" % args.tag
            curve = [{"sweep": 0, "ll_per_token": (s.model_log_likelihood() / ntok).tolist()}]
            ms = 0.0
            for it in range(1, args.sweeps + 1):
                if flags & SWEEP_REUSE_TREES:
                    s.build_trees()
                st = s.sweep(it, seed, flags=flags)
                ms += st.total_ms
                if it % args.every == 0 or it == args.sweeps:
                    curve.append({"sweep": it, "ll_per_token": (s.model_log_likelihood() / ntok).tolist(),
                                  "changed_frac": st.changed / max(1, st.tokens), "ms_per_sweep": ms / it})
            print(f"{name}: final LL/token {curve[-1]['ll_per_token']}  {ms / args.sweeps:.2f} ms/sweep", flush=True)
            runs[name] = curve
            s.close()
    json.dump({"runs": runs, "workload": args.workload, "docs": c.D, "tokens": c.total_tokens}, open(args.out, "w"), indent=1)


def run_group(args):
    """LL curve of a live chain over N document shards on ONE device, synchronous exchange against MVHDP_SWEEP_ASYNC_EXCHANGE."""
    from mvtopicmodel_amd import NativeGroup, NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import SWEEP_ASYNC_EXCHANGE, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS
    c, hy, K_init = load(args.workload, args.docs)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    ntok = np.array([int(c.doc_off[m][-1]) for m in range(c.M)], dtype=np.float64)
    tot = sum(np.diff(c.doc_off[m]) for m in range(c.M))
    runs = {}
    for asyn in ((0,) if args.no_async else (0, 1)):          # (ASYNC_EXCHANGE is refused while a topic is inactive: a truncated HDP runs --no-async)
        shards = []
        for lo, hi in synth.shard_bounds(tot, args.shards):
            sub = c.slice_docs(lo, hi)
            s = NativeSampler(c.K, c.V, doc_id_base=lo)
            for m in range(c.M):
                s.set_corpus(m, sub.doc_off[m], sub.tokens[m]); s.set_assignments(m, z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
            s.set_hyper(hy); s.build_counts()
            shards.append(s)
        g = NativeGroup(shards)
        g.build_counts()
        flags = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(args.segments) | (SWEEP_ASYNC_EXCHANGE if asyn else 0)
        name = f"gpu live over {args.shards} shards, {args.segments} segments, " + ("exchange one sweep behind (ASYNC_EXCHANGE)" if asyn else "exchange after every sweep")
        curve = [{"sweep": 0, "ll_per_token": (g.model_log_likelihood() / ntok).tolist()}]
        for it in range(1, args.sweeps + 1):
            g.sweep(it, args.seed, flags=flags)
            if it % args.every == 0 or it == args.sweeps:
                curve.append({"sweep": it, "ll_per_token": (g.model_log_likelihood() / ntok).tolist()})
        print(f"{name}: final LL/token {curve[-1]['ll_per_token']}", flush=True)
        runs[name] = curve
        g.close()
        for s in shards:
            s.close()
    json.dump({"runs": runs, "workload": args.workload, "docs": c.D, "tokens": c.total_tokens}, open(args.out, "w"), indent=1)


def table(args):
    runs = {}
    meta = None
    for f in args.files:
        j = json.load(open(f))
        runs.update(j["runs"])
        meta = meta or j
    names = list(runs)
    sweeps = sorted({p["sweep"] for r in runs.values() for p in r})
    M = len(next(iter(runs.values()))[0]["ll_per_token"])
    print(f"# LL/token against the number of sweeps: {meta['workload']}, {meta['docs']} entities, {meta['tokens']} tokens\n")
    print("modelLogLikelihood (PTM:3322-3452) divided by the view's token count; same corpus, same initial assignments, "
          "same fixed hyper-parameters for every run (tools/ll_curves.py).\n")
    for m in range(M):
        print(f"## view {m}\n")
        print("| sweeps | " + " | ".join(names) + " |")
        print("|---|" + "---|" * len(names))
        for sw in sweeps:
            row = []
            for n in names:
                v = [p["ll_per_token"][m] for p in runs[n] if p["sweep"] == sw]
                row.append(f"{v[0]:.4f}" if v else "")
            print(f"| {sw} | " + " | ".join(row) + " |")
        print()
    # how many sweeps of each run reach the LL the cpu run has after n sweeps (view 0)
    cpu = [n for n in names if n.startswith("cpu")]
    if cpu:
        ref = runs[cpu[0]]
        print("## sweeps needed to reach the LL/token of the CPU run (view 0, linear interpolation between samples)\n")
        others = [n for n in names if n not in cpu]
        print("| cpu sweeps | cpu LL/token | " + " | ".join(others) + " |")
        print("|---|---|" + "---|" * len(others))
        for p in ref:
            if p["sweep"] == 0:
                continue
            target = p["ll_per_token"][0]
            row = []
            for n in others:
                xs = [q["sweep"] for q in runs[n]]
                ys = [q["ll_per_token"][0] for q in runs[n]]
                hit = ""
                for i in range(1, len(xs)):
                    if ys[i] >= target > ys[i - 1]:
                        hit = f"{xs[i - 1] + (target - ys[i - 1]) / (ys[i] - ys[i - 1]) * (xs[i] - xs[i - 1]):.1f}"
                        break
                if not hit:
                    hit = "0" if ys[0] >= target else f">{xs[-1]}"
                row.append(hit)
            print(f"| {p['sweep']} | {target:.4f} | " + " | ".join(row) + " |")


def _sweeps_needed(curve, m, target):
    """GPU sweeps at which view m of `curve` first reaches `target` (linear interpolation between samples); None: not within the run"""
    xs = [q["sweep"] for q in curve]
    ys = [q["ll_per_token"][m] for q in curve]
    if ys[0] >= target:
        return 0.0
    for i in range(1, len(xs)):
        if ys[i] >= target > ys[i - 1]:
            return xs[i - 1] + (target - ys[i - 1]) / (ys[i] - ys[i - 1]) * (xs[i] - xs[i - 1])
    return None


def _mode_of(name):
    """run name without its seed / tag: the update mode"""
    import re
    return re.sub(r"\s*(seed \d+|\[[^\]]*\])", "", name).strip()


def equivalents(args):
    """What a sweep of each GPU update mode is worth IN EVERY VIEW: GPU sweeps needed to reach the LL/token the CPU restatement of the
    reference reaches after n sweeps, n = 10, 20, ..., against every CPU chain given (two chains = the band of a nondeterministic
    reference) and every seed of the mode; per view the range over all of them, and the worst view."""
    runs, meta = {}, None
    for f in args.files:
        j = json.load(open(f))
        runs.update(j["runs"]); meta = meta or j
    cpus = {n: r for n, r in runs.items() if n.startswith("cpu")}
    M = len(next(iter(runs.values()))[0]["ll_per_token"])
    out = {"workload": meta["workload"], "docs": meta["docs"], "cpu_chains": sorted(cpus), "cpu_sweeps": [], "modes": {}, "sources": [os.path.basename(f) for f in args.files]}
    targets = sorted({p["sweep"] for r in cpus.values() for p in r if p["sweep"] >= args.first and p["sweep"] % args.step == 0 and p["sweep"] <= args.last})
    out["cpu_sweeps"] = targets
    # the band of the reference itself: CPU chain A's sweeps needed to reach chain B's LL
    band = {}
    names = sorted(cpus)
    for m in range(M):
        rs = []
        for a in names:
            for b in names:
                if a == b:
                    continue
                for t in targets:
                    tv = [p["ll_per_token"][m] for p in cpus[b] if p["sweep"] == t]
                    if tv:
                        n = _sweeps_needed(cpus[a], m, tv[0])
                        if n is not None:
                            rs.append(n / t)
        band[f"view{m}"] = [min(rs), max(rs)] if rs else None
    out["cpu_band"] = band
    modes = {}
    for n, r in runs.items():
        if not n.startswith("cpu"):
            modes.setdefault(_mode_of(n), []).append(r)
    for mode, rr in modes.items():
        per_view, worst = {}, None
        for m in range(M):
            ratios, missed = [], 0
            for r in rr:
                for cn in names:
                    for t in targets:
                        tv = [p["ll_per_token"][m] for p in cpus[cn] if p["sweep"] == t]
                        if not tv:
                            continue
                        n = _sweeps_needed(r, m, tv[0])
                        if n is None:
                            missed += 1
                        else:
                            ratios.append(n / t)
            per_view[f"view{m}"] = {"min": min(ratios) if ratios else None, "max": max(ratios) if ratios else None, "not_reached": missed,
                                    "n": len(ratios) + missed}
            if ratios and (worst is None or max(ratios) > worst[1] or missed):
                worst = (m, max(ratios), missed)
        modes[mode] = {"runs": len(rr), "per_view": per_view,
                       "worst_view": None if worst is None else {"view": worst[0], "max": worst[1], "not_reached": worst[2]}}
    out["modes"] = modes
    json.dump(out, open(args.out, "w"), indent=1)
    print(f"# GPU sweeps per sweep of the CPU restatement of the reference, per view: {out['workload']}, {out['docs']} entities; CPU sweeps {targets[0]}..{targets[-1]}\n")
    print("the reference's own band (one CPU chain against the other): " + ", ".join(f"view {m}: {band[f'view{m}'][0]:.2f}-{band[f'view{m}'][1]:.2f}" for m in range(M) if band[f"view{m}"]) + "\n")
    print("| mode | runs | " + " | ".join(f"view {m}" for m in range(M)) + " |")
    print("|---|---|" + "---|" * M)
    for mode, d in modes.items():
        cells = []
        for m in range(M):
            v = d["per_view"][f"view{m}"]
            cells.append("not reached" if v["min"] is None else f"{v['min']:.2f}-{v['max']:.2f}" + (f" ({v['not_reached']}/{v['n']} not reached)" if v["not_reached"] else ""))
        print(f"| {mode} | {d['runs']} | " + " | ".join(cells) + " |")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("cpu", "gpu"):
        p = sub.add_parser(name)
        p.add_argument("--workload", default="C3")
        p.add_argument("--docs", type=int, default=200000)
        p.add_argument("--sweeps", type=int, default=100)
        p.add_argument("--every", type=int, default=5)
        p.add_argument("--seed", type=int, default=20260101)
        p.add_argument("--out", required=True)
        if name == "cpu":
            p.add_argument("--threads", type=int, default=8)
        else:
            p.add_argument("--live-segments", type=int, nargs="*", default=[1, 4, 16])
            p.add_argument("--segmented", type=int, nargs="*", default=[], help="also run SEGMENT_APPLY sweeps with these segment counts")
            p.add_argument("--overlapped", type=int, nargs="*", default=[], help="also run SEGMENT_APPLY | SEGMENT_OVERLAP sweeps with these segment counts")
            p.add_argument("--live-trees-once", type=int, nargs="*", default=[], help="also run live sweeps of n segments whose trees are built once per sweep")
            p.add_argument("--only", nargs="*", default=[], help="keep only the modes whose name contains one of these strings")
            p.add_argument("--live16", type=int, nargs="*", default=[], help="pin mvhdp_tuning.live16 (1: live sweeps keep the light n_wk rows in the 16-bit mirror); several values = one run each")
            p.add_argument("--seeds", type=int, nargs="*", default=[], help="one run per seed of every mode (the noise band of a chain)")
            p.add_argument("--tag", default="", help="appended to the run names (which library build this was)")
            p.add_argument("--live-rows", type=int, default=None, help="pin mvhdp_tuning.live_rows: 1 the tree branch of a live sweep samples from the live count rows, 0 stored trees rebuilt at every segment border")
    p = sub.add_parser("group")
    p.add_argument("--workload", default="C3"); p.add_argument("--docs", type=int, default=200000)
    p.add_argument("--sweeps", type=int, default=100); p.add_argument("--every", type=int, default=5)
    p.add_argument("--seed", type=int, default=20260101); p.add_argument("--out", required=True)
    p.add_argument("--shards", type=int, default=8); p.add_argument("--segments", type=int, default=4)
    p.add_argument("--no-async", action="store_true", help="only the synchronous exchange")
    p = sub.add_parser("table")
    p.add_argument("files", nargs="+")
    p = sub.add_parser("equivalents")
    p.add_argument("files", nargs="+")
    p.add_argument("--out", required=True)
    p.add_argument("--first", type=int, default=10); p.add_argument("--last", type=int, default=100); p.add_argument("--step", type=int, default=10)
    args = ap.parse_args()
    {"cpu": run_cpu, "gpu": run_gpu, "group": run_group, "table": table, "equivalents": equivalents}[args.cmd](args)


if __name__ == "__main__":
    main()
