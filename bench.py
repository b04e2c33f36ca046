#!/usr/bin/env python3
"""bench.py — Gibbs tokens sampled per second per sweep (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload C4] [--docs D]

A "step" is one full Gibbs sweep (mvhdp_sweep: view weights + F+tree rebuild +
the sweep kernel + count update) over the synthetic corpus, inputs resident in
HBM before the timed region.  N=1 runs BASELINE config C4 (1M entities x 3 views,
K=400, ~150M tokens: the configuration the metric is quoted on; it fits one
MI355X).  N>1 (launched by torch.distributed.run, one rank per GPU) shards the
same corpus by token count and adds the per-sweep RCCL all-reduce of the count
deltas: total work is fixed, so "scaling" is "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_token(K):
    """one int32 n_wk row + token id + old assignment."""
    return 4 * K + 8


def cpu_baseline(workload, sample_docs, iters, seed):
    """The reference's thread topology restated in C (oracle/ref_threaded.c), timed on the
    host cores of this box on a bounded sample of the same workload.  Reported, not a target."""
    from oracle.binding import Oracle
    from mvtopicmodel_amd import synth
    from mvtopicmodel_amd.native import Hyper
    c = synth.make_config(workload, doc_lo=0, doc_hi=sample_docs)
    hy = Hyper.defaults(c.K, c.V)
    o = Oracle(c.K, c.V)
    for m in range(c.M):
        o.set_corpus(m, c.doc_off[m], c.tokens[m])
    o.set_hyper(hy.alpha, hy.alpha_sum, hy.beta, hy.beta_sum, hy.gamma, hy.p_a, hy.p_b, None)
    o.init_assignments(1)
    o.build_counts()
    cores = os.cpu_count() or 4
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    T = max(4, min(cores, 16))     # the GPU box's CPU share for one GPU is 16 cores
    secs, st = o.threaded_estimate(T, iters, seed)
    o.close()
    return {"value": st["tokens"] / secs, "unit": "tokens/s", "cores": T, "kind": "port",
            "sample": f"first {sample_docs} entities of {workload} ({c.total_tokens} tokens) x {iters} iterations, "
                      f"{3 * T // 4} sampler + {T // 4} updater threads (PTM:1036-1037 topology)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--docs", type=int, default=None, help="override the entity count (smoke runs)")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-docs", type=int, default=120000)
    ap.add_argument("--live", action="store_true",
                    help="MVHDP_SWEEP_LIVE: atomics straight on the shared counts (the reference's update discipline); "
                         "not bit-reproducible, so not the default")
    ap.add_argument("--live-segments", type=int, default=0, help="F+tree rebuilds per live sweep (0 = library default)")
    ap.add_argument("--live-steps", type=int, default=None,
                    help="live sweeps timed after the K steps for the 'live' object (default: 5 on one GPU, 0 on several; 0 = none)")
    ap.add_argument("--plain-exchange", action="store_true", help="N>1: one all-reduce then apply, instead of the chunked pipeline")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 ranks all on cuda:0 with the gloo backend (host-staged all-reduce): exercises the "
                         "sharding logic on a 1-GPU box; not a performance number")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the
            # GPU (torch is not even imported yet); the ranks are children of torch.distributed.run, rank 0
            # prints the JSON line on the inherited stdout, and this process exits with the launcher's code.
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd))
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.dist import GpuShard, build_counts_all_reduce, sweep_all_reduce
    from mvtopicmodel_amd.host import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sweep has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(device))

    cfg = dict(synth.CONFIGS[args.workload])
    D_total = args.docs or cfg["D"]
    K, V = cfg["K"], cfg["V"]
    M = len(V)

    # ---- document shard of this rank, balanced by token count ----
    t_setup = time.time()
    doc_tokens = synth.config_doc_token_counts(args.workload, D=D_total)
    lo, hi = synth.shard_bounds(doc_tokens, world)[rank]
    corpus = synth.make_config(args.workload, D=D_total, doc_lo=lo, doc_hi=hi)
    total_tokens = int(doc_tokens.sum())
    # initial assignments: the addInstances draw order over the whole corpus (it depends on
    # entity lengths only), of which this rank keeps its slice
    inactive, K_init = synth.config_inactive(args.workload)
    if lo == 0:
        z0 = init_assignments(K_init, corpus.doc_off, seed=1)
    else:
        lens = synth._doc_lengths(V, 0, hi, cfg["lam"], cfg["seed"], [1.0] + [0.8] * (M - 1), cfg.get("power_law_text", False))
        offs = [np.concatenate([[0], np.cumsum(L)]).astype(np.int64) for L in lens]
        zfull = init_assignments(K_init, offs, seed=1)
        z0 = [zfull[m][offs[m][lo]:offs[m][hi]] for m in range(M)]
        del zfull, lens, offs

    s = NativeSampler(K, V, device=local_rank, doc_id_base=lo)
    for m in range(M):
        s.set_corpus(m, corpus.doc_off[m], corpus.tokens[m])
        s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(K, V, inactive=inactive))
    sweep_flags = 0
    if args.live:
        sweep_flags = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(args.live_segments)
    shard = GpuShard(s, device, host_staged=args.rehearse_on_one_gpu)
    build_counts_all_reduce(shard)
    local_tokens = corpus.total_tokens
    del z0
    setup_s = time.time() - t_setup

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the exchange step: pipelined (chunked all-reduce overlapped with apply + tree rebuild) unless it fails in the warm-up,
    # in which case every rank falls back to the plain sequence (sweep -> one all-reduce -> apply) and the line says so
    exchange = {"pipeline": not args.plain_exchange}
    for w in range(args.warmup):
        try:
            sweep_all_reduce(shard, w, args.seed, flags=sweep_flags, pipeline=exchange["pipeline"])
        except Exception as e:
            if not exchange["pipeline"] or world == 1:
                raise
            exchange = {"pipeline": False, "fallback_reason": repr(e)}
            for cleanup in (lambda: s.apply_delta_end(-1, -1), lambda: s.apply_delta(-1, -1), s.synchronize):
                try:                                 # close a half-open apply bracket, drop pending deltas
                    cleanup()
                except Exception:
                    pass
            sweep_all_reduce(shard, w, args.seed, flags=sweep_flags, pipeline=False)
    barrier()
    kernel_ms = []
    phases = {}
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = sweep_all_reduce(shard, args.warmup + k, args.seed, flags=sweep_flags, timings=phases, pipeline=exchange["pipeline"])
        kernel_ms.append(last.sweep_kernel_ms)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    value = total_tokens * args.steps / dt
    # roofline of the dominant kernel (sweep_fast_kernel<R>) on this rank: algorithmic bytes per launch /
    # its average duration, measured with hipEvents on the library's stream
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    bpt = algorithmic_bytes_per_token(K)
    achieved = local_tokens * bpt / avg_kernel_s / 1e9
    # HBM-side traffic of the same kernels from the committed PMC passes (profiles/pmc_r02b.sh: separate FETCH_SIZE /
    # TCC_EA0_RDREQ / WRITE_SIZE runs of this very command); bytes per token there x tokens per launch here.
    traffic = None
    traffic_note = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r02b_c4_pmc_summary.json")))
        if args.workload == "C4":
            bpt_meas = pm["fetch_bytes_per_token"] + pm["write_bytes_per_token"]
            traffic = local_tokens * bpt_meas / avg_kernel_s / 1e9
            traffic_note = (f"{bpt_meas:.0f} B/token = TCC_EA0_RDREQ x 128 B (= 2 x FETCH_SIZE: the gfx950 correction, calibrated "
                            "on this access pattern in profiles/r02_fetch_calibration.txt) + WRITE_SIZE, Infinity-Cache hits "
                            "included; PMC passes of profiles/pmc_r02b.sh over the 20 timed sweeps of `--steps 20 --warmup 5` "
                            "(a rocprofv3 run of its own; the walk threshold of the chunk head moves during those sweeps and the "
                            "1-round kernel with its 16-bit mirror of the counts takes over at sweep 13, so the per-token figure "
                            "belongs to that window and is BELOW the algorithmic 4K+8 bytes, which assume 4-byte counts), the rate is this run's")
    except Exception:
        pass
    out = {
        "metric": "gibbs_tokens_per_sec", "value": value, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if args.rehearse_on_one_gpu else ""),
        "config": {"workload": f"{args.workload}: {D_total} entities x {M} views, K={K}, vocab {V}, "
                               f"{total_tokens} tokens; doc-sharded across {world} GPU(s)",
                   "topics": K, "views": M, "tokens": total_tokens, "entities": D_total,
                   "sharding": f"documents/{world}, per-sweep int32 all-reduce of n_wk,n_k deltas" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel": "sweep kernels of one mvhdp_sweep (dominant: sweep_fast_kernel<R>)", "bytes_per_token": bpt, "tokens_per_launch": local_tokens,
                     "avg_kernel_ms": avg_kernel_s * 1e3},
        "sweep": {"changed_frac": last.changed / max(1, last.tokens),
                  "branch_frac": {"new": last.new_mass_cnt / max(1, last.tokens),
                                  "doc": last.topic_doc_mass_cnt / max(1, last.tokens),
                                  "tree": last.word_ftree_mass_cnt / max(1, last.tokens)},
                  "exact_fallbacks": last.exact_fallbacks, "total_ms_last": last.total_ms},
        "setup_s": setup_s,
        # rank 0's milliseconds per step by phase (mvtopicmodel_amd.dist.sweep_all_reduce): host wall time of the
        # sweep call (view weights, F+tree rebuild, kernels, statistics read-back), of which device time in the sweep
        # kernels; the collective (device time by events on the shared stream); apply (waits for the collective)
        "phase_ms": {k: v / max(1, phases.get("n", 1)) for k, v in phases.items() if k != "n"},
        "exchange": ("none (one shard)" if world == 1 else
                     ("pipelined: %d row-range all-reduces overlapped with apply + tree rebuild" % 4 if exchange["pipeline"] else
                      "plain: one all-reduce, then apply" + (" (pipeline failed in warm-up: %s)" % exchange["fallback_reason"] if "fallback_reason" in exchange else ""))),
        "update_mode": ("live, %d tree rebuilds per sweep" % (args.live_segments or 4)) if args.live else "deferred (snapshot sweep, bit-reproducible)",
    }
    # order-independent fingerprint of the final global counts: must not depend on the number of shards
    nk_fp = [int(np.asarray(s.get_counts(m)[1], dtype=np.int64).dot(np.arange(1, K + 1, dtype=np.int64))) for m in range(M)]
    out["final_nk_fingerprint"] = nk_fp
    if args.live_steps is None:
        args.live_steps = 5 if world == 1 else 0
    if not args.live and args.live_steps > 0:
        # The other update mode, timed after (and outside) the K steps above: MVHDP_SWEEP_LIVE, the reference's own
        # discipline.  profiles/r02_ll_curves.md: a live sweep is worth one sweep of the CPU reference (0.93-1.0 sweeps
        # needed per reference sweep on C3), a deferred sweep 0.4-0.7 of one -- quote tokens/s with that in mind.
        lf = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(args.live_segments)
        try:
            sweep_all_reduce(shard, args.warmup + args.steps, args.seed, flags=lf)
            barrier()
            t1 = time.perf_counter()
            for k in range(args.live_steps):
                sweep_all_reduce(shard, args.warmup + args.steps + 1 + k, args.seed, flags=lf)
            barrier()
            dl = time.perf_counter() - t1
            if world > 1:
                tl = torch.tensor([dl], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
                dist.all_reduce(tl, op=dist.ReduceOp.MAX)
                dl = float(tl.item())
        except Exception as e:                      # the secondary measurement must never cost the primary one
            out["live"] = {"error": repr(e)}
            dl = None
        if dl is not None:
            out["live"] = {"value": total_tokens * args.live_steps / dl, "unit": "tokens/s", "steps": args.live_steps,
                           "ms_per_step": dl / args.live_steps * 1e3, "tree_rebuilds_per_sweep": args.live_segments or 4,
                           "note": "MVHDP_SWEEP_LIVE (atomics on the shared n_wk, UPD:197-207), timed after the K deferred steps; "
                                   "sweep-for-sweep equal to the CPU reference (profiles/r02_ll_curves.md), not bit-reproducible"}
    if world == 1 and not args.live and args.live_steps > 0:
        # the deterministic middle ground, single GPU only: a deferred sweep in 8 interleaved segments with the deltas
        # applied in between (MVHDP_SWEEP_SEGMENT_APPLY): bit-exact against the oracle like the plain deferred sweep and
        # 0.94-1.07 sweeps per sweep of the CPU reference (profiles/r02_ll_curves.md)
        try:
            from mvtopicmodel_amd.native import SWEEP_SEGMENT_APPLY
            sf = SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(8)
            base = args.warmup + args.steps + args.live_steps + 1
            s.sweep(base, args.seed, flags=sf)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.live_steps):
                s.sweep(base + 1 + k, args.seed, flags=sf)
            torch.cuda.synchronize()
            dsg = time.perf_counter() - t1
            out["segmented"] = {"value": total_tokens * args.live_steps / dsg, "unit": "tokens/s", "steps": args.live_steps,
                                "ms_per_step": dsg / args.live_steps * 1e3, "segments": 8,
                                "note": "MVHDP_SWEEP_SEGMENT_APPLY: deferred sweep in 8 segments, deltas applied and trees rebuilt in "
                                        "between; deterministic (oracle-checked), about one reference sweep per sweep"}
        except Exception as e:
            out["segmented"] = {"error": repr(e)}
    shard.close()
    s.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, min(args.cpu_sample_docs, D_total), 3, args.seed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
