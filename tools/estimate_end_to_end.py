#!/usr/bin/env python3
"""The whole caller path, end to end, through the host mirror of FastQMVWVParallelTopicModel (the C++ stand-in for the Java
class): addInstances (name alignment, java.util.Random initial topics, counts, trees) and estimate() with its schedule --
burn-in p_a ramp, then optimizeP / optimizeDP / optimizeGamma / optimizeBeta every optimizeInterval iterations, LL/token
every 10 (PTM:1146-1320) -- on a BASELINE config, one GPU.  Reports where the wall time goes: plain iterations (one
mvhdp_sweep each), iterations that also optimise, iterations that also compute the log-likelihood.

  python tools/estimate_end_to_end.py --workload C3 --iterations 250 --burnin 100 --optimize-interval 50 [--live]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--docs", type=int, default=None)
    ap.add_argument("--iterations", type=int, default=250)
    ap.add_argument("--burnin", type=int, default=100)
    ap.add_argument("--optimize-interval", type=int, default=50)
    ap.add_argument("--live", action="store_true")
    ap.add_argument("--device-gamma", action="store_true", help="optimizeGamma's per-entity sums on the device")
    ap.add_argument("--device-tables", action="store_true", help="optimizeDP's view-table simulation (the Antoniak draws) on the device")
    ap.add_argument("--shards", type=int, default=1, help="keep the model as n document shards behind an mvhdp_group (all on this GPU)")
    args = ap.parse_args()
    from mvtopicmodel_amd import synth
    from hostmirror.binding import FastQMVWVParallelTopicModel
    cfg = synth.CONFIGS[args.workload]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(args.workload, D=args.docs)
    # MALLET InstanceLists: a view lists only the entities that have it (names = entity ids)
    training = []
    for m in range(M):
        lens = np.diff(c.doc_off[m])
        have = np.flatnonzero(lens > 0)
        off = np.concatenate([[0], np.cumsum(lens[have])]).astype(np.int64)
        training.append((have.astype(np.int64), off, c.tokens[m], V[m]))
    model = FastQMVWVParallelTopicModel(K, M, 0.1, 0.01)
    model.setNumIterations(args.iterations); model.setBurninPeriod(args.burnin)
    model.setOptimizeInterval(args.optimize_interval); model.setRandomSeed(1)
    model.setLiveUpdates(args.live)
    model.setDeviceGammaStatistics(args.device_gamma)
    model.setDeviceTableStatistics(args.device_tables)
    model.setNumShards(args.shards)
    t0 = time.perf_counter()
    model.addInstances(training)
    t_add = time.perf_counter() - t0
    t0 = time.perf_counter()
    model.estimate()
    t_est = time.perf_counter() - t0
    log = model.iteration_log()
    ms = np.array([x[0] for x in log])
    it = np.arange(1, len(ms) + 1)
    opt = (it > args.burnin) & (it % args.optimize_interval == 0)
    ll = (it % 10 == 0)
    tw = (it % 50 == 0)                          # displayTopWords every showTopicsInterval = 50 iterations (PTM:117,1150-1152): the reference's logging, on the host
    plain = ~opt & ~ll & ~tw
    kern = np.array([x[1]["sweep_kernel_ms"] for x in log])
    out = {
        "workload": args.workload, "entities": model.num_entities(), "tokens": c.total_tokens, "update_mode": "live" if args.live else "deferred", "optimizeGamma_document_sums": "device" if args.device_gamma else "host loop (reference)", "optimizeDP_view_tables": "device" if args.device_tables else "host loop (reference)", "shards": args.shards,
        "iterations": args.iterations, "burnin": args.burnin, "optimize_interval": args.optimize_interval,
        "addInstances_s": round(t_add, 3), "estimate_s": round(t_est, 3),
        "ms_per_plain_iteration_median": round(float(np.median(ms[plain])), 3),
        "sweep_kernel_ms_median": round(float(np.median(kern)), 3),
        "ms_per_iteration_with_LL_median": round(float(np.median(ms[ll & ~opt & ~tw])), 3),
        "ms_per_optimising_iteration_median": round(float(np.median(ms[opt & ~tw])), 3) if (opt & ~tw).any() else None,
        "ms_per_iteration_with_top_words_median": round(float(np.median(ms[tw & ~opt])), 3) if (tw & ~opt).any() else None,
        "optimising_iterations": int(opt.sum()),
        "share_of_estimate_time": {"sweeps": round(float(kern.sum() / 1e3 / t_est), 3),
                                   "optimise_extra": round(float((ms[opt & ~tw] - np.median(ms[plain])).sum() / 1e3 / t_est), 3) if (opt & ~tw).any() else 0.0,
                                   "top_words_extra": round(float((ms[tw & ~opt] - np.median(ms[plain])).sum() / 1e3 / t_est), 3) if (tw & ~opt).any() else 0.0,
                                   "log_likelihood_extra": round(float((ms[ll & ~opt & ~tw] - np.median(ms[plain])).sum() / 1e3 / t_est), 3)},
        # between two optimisations past the burn-in: the sweeps' share of the interval's wall time (logging apart)
        "sweeps_share_of_an_optimise_interval": round(float(args.optimize_interval * np.median(ms[plain]) / (args.optimize_interval * np.median(ms[plain]) + (np.median(ms[opt & ~tw]) - np.median(ms[plain])))), 3) if (opt & ~tw).any() else None,
        "tokens_per_s_over_estimate": round(c.total_tokens * args.iterations / t_est / 1e9, 3),
        "LL_per_token_view0": [round(float(x), 4) for x in model.perplexities(0)[1:]][:: max(1, args.iterations // 100)],
    }
    print(json.dumps(out))
    model.close()


if __name__ == "__main__":
    main()
