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


SIMDS = 256 * 4           # CUs x SIMDs per CU
SIMD_CLOCK_GHZ = 2.4      # MI355X peak engine clock


def physical_rooflines(workload, tokens_per_launch, avg_kernel_s, mode="deferred"):
    """The two ceilings that can actually bind the sweep kernels, as fractions below 1 (the fixed-byte yardstick of SURVEY 8d assumes
    4-byte counts and a whole row per token; the kernels gather 2-byte cells, so that fraction passes 1):

      physical  fabric bytes per token -- (TCC_EA0_RDREQ x 128 B + WRITE_SIZE), the PMC passes of profiles/profile_r05.sh over the very
                window this command times -- x tokens per launch / the kernel time measured HERE, against the 8 TB/s HBM peak and
                against what a bare gather of the same rows reaches on this chip (tools/microbench/gather_patterns.hip)
      issue     cycles per token during which a SIMD's vector ALU (scalar unit) is busy -- SQ_ACTIVE_INST_VALU / _SCA x 4 of the SQ
                pass of the same recipe -- x tokens per second here, against 1024 SIMDs x 2.4 GHz

    Counters cannot be read inside an un-profiled run: the per-token figures are the committed profile's (same build, same command),
    the rate is this run's.  Returns (physical, issue) dicts or (None, None) when the profile of this workload is not committed."""
    try:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r05_roofline_inputs.json")))
            w = prof["workloads"][workload][mode]
        except Exception:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r04_roofline_inputs.json")))     # (a workload profiled in round 4 only: the deferred kernels are unchanged)
            w = prof["workloads"][workload][mode]
    except Exception:
        return None, None
    tok_s = tokens_per_launch / avg_kernel_s
    bpt = w["fabric_read_bytes_per_token"] + w["write_bytes_per_token"]
    gbs = tok_s * bpt / 1e9
    # the bare gather of THIS workload's rows (mirror row length, typical list size), where it was measured
    ceil = (prof.get("gather_ceilings_GBs") or {}).get(workload)
    ceil_src = (prof.get("gather_ceilings_source") or {}).get(workload)
    if ceil is None and prof.get("gather_ceiling_workload", workload) == workload:
        ceil, ceil_src = prof.get("gather_ceiling_GBs"), prof.get("gather_ceiling_source")
    physical = {"bytes_per_token": bpt, "achieved": gbs, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                "gather_ceiling": ceil, "frac_of_gather_ceiling": (gbs / ceil) if ceil else None,
                "source": w.get("pmc_source"), "gather_ceiling_source": ceil_src,
                # the cross-check: this run's kernel time (roofline.avg_kernel_ms, hipEvents) against the profiled runs' the counters are from
                "profiled_kernel_ms_per_sweep": w.get("kernel_ms_per_sweep_profiled")}
    issue = None
    if "valu_busy_cycles_per_token" in w:
        cap = SIMDS * SIMD_CLOCK_GHZ * 1e9
        issue = {"valu_busy_cycles_per_token": w["valu_busy_cycles_per_token"], "scalar_busy_cycles_per_token": w["scalar_busy_cycles_per_token"],
                 "valu_insts_per_token": w.get("valu_insts_per_token"), "scalar_insts_per_token": w.get("scalar_insts_per_token"),
                 "simd_cycles_per_s": cap, "frac": tok_s * w["valu_busy_cycles_per_token"] / cap,
                 "frac_scalar": tok_s * w["scalar_busy_cycles_per_token"] / cap, "source": w.get("sq_source")}
    return physical, issue


def sweep_equivalents(K, workload=None):
    """GPU sweeps per sweep of the CPU restatement of the reference, PER VIEW, by update mode: how many sweeps of the mode reach the
    LL/token the reference reaches after n sweeps (n = 10 .. 100), against two chains of the (nondeterministic) reference and every seed
    of the mode -- tools/ll_curves.py equivalents, committed as profiles/r05_sweep_equivalents_*.json.  `worst_view` is the range of the
    view that needs most; `cpu_band` is the reference against itself (one chain's sweeps to reach the other's LL)."""
    name = "r05_sweep_equivalents_c5_100k.json" if workload == "C5" else "r05_sweep_equivalents_c2.json" if workload == "C2" else "r05_sweep_equivalents_c4_200k.json" if K >= 256 else "r05_sweep_equivalents_c3.json"
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception as e:
        return {"source": f"profiles/{name} missing ({e!r})", "gpu_sweeps_per_reference_sweep": {}, "per_view": {}}
    key = {"deferred": "gpu deferred (snapshot sweep)", "live": "gpu live, default segment(s) per sweep",
           "segmented": "gpu deferred in 8 segments, applied in between", "live_stored_trees_4_segments": "gpu live, 4 segment(s) per sweep (stored trees rebuilt per segment)"}
    out, per_view = {}, {}
    for mode, run in key.items():
        d = j["modes"].get(run)
        if not d:
            continue
        pv = {v: ([round(x["min"], 3), round(x["max"], 3)] if x["min"] is not None else None) for v, x in d["per_view"].items()}
        nr = {v: f"{x['not_reached']}/{x['n']}" for v, x in d["per_view"].items() if x["not_reached"]}
        per_view[mode] = {"range": pv, "not_reached": nr}
        worst = max((x["max"] for x in d["per_view"].values() if x["max"] is not None), default=None)
        best = min((x["min"] for x in d["per_view"].values() if x["min"] is not None), default=None)
        out[mode] = [round(best, 3), round(worst, 3)] if worst is not None else None
    return {"source": f"profiles/{name} ({j['workload']}, {j['docs']} entities; CPU sweeps {j['cpu_sweeps'][0]}..{j['cpu_sweeps'][-1]}; sources {', '.join(j['sources'])})",
            "gpu_sweeps_per_reference_sweep": out,           # over ALL views: [fewest, most]
            "per_view": per_view, "cpu_band": j.get("cpu_band"),
            "note": "ranges over two CPU chains x the mode's seeds x CPU sweeps 10..100; 'not_reached': comparisons in which the mode's run ended below the reference's LL"}


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
                    help="sweeps of each secondary update mode timed after the K steps, behind 3 warm-up sweeps of that mode "
                         "(default: 10 on one GPU, 0 on several; 0 = none)")
    ap.add_argument("--batch", action="store_true",
                    help="one GPU: the K timed steps as ONE mvhdp_sweep_many call (no host round trip between sweeps): what a host "
                         "does when nothing needs the counts in between; same integers")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="N>1: the exchange through torch.distributed on host copies (the fallback path) instead of the library's own RCCL group")
    ap.add_argument("--rehearse-native", action="store_true",
                    help="N>1 ranks all on cuda:0 but WITH the library's own RCCL group: RCCL refuses two ranks on one GPU, so this "
                         "exercises the id broadcast, the failure agreement across ranks and the fallback path end to end")
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
    from mvtopicmodel_amd import NativeGroup, NativeSampler, synth
    from mvtopicmodel_amd.dist import GpuShard, build_counts_all_reduce, sweep_all_reduce
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE, SWEEP_LIVE_SEGMENTS, SWEEP_SEGMENT_APPLY

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sweep has no CPU fallback")
    if args.rehearse_on_one_gpu or args.rehearse_native:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        # torch.distributed is the launcher's control plane only (gloo on the host: the 128-byte RCCL id, barriers, the max over
        # ranks of the wall time); the data path -- the all-reduce of the count deltas -- is RCCL inside libmvhdp.so (mvhdp_group_*)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")

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
    local_tokens = corpus.total_tokens
    del z0

    def all_ranks_ok(ok):
        """Every rank learns whether EVERY rank succeeded (a failure on one rank must not leave the others inside a collective)."""
        if world == 1:
            return ok
        t = torch.tensor([0 if ok else 1], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) == 0

    # ---- the exchange step ----
    #   world == 1            none: one handle, the library applies its own deltas
    #   world  > 1            mvhdp_group_* (RCCL inside the library), one rank per process
    #   fallback              the same shards through torch.distributed on host copies (gloo): taken by ALL ranks together when the
    #                         native group cannot be formed or fails in the warm-up on ANY rank; the model is recounted from z
    #                         first (mvhdp_build_counts drops pending deltas), so the numbers that follow are of a consistent model
    group = shard = None
    exchange = {"kind": "none (one shard)"}
    if world > 1 and not args.rehearse_on_one_gpu and not args.torch_exchange:
        err = None
        try:
            ids = [NativeGroup.unique_id() if rank == 0 else None]
        except Exception as e:
            ids, err = [None], repr(e)
        dist.broadcast_object_list(ids, src=0)
        try:
            if ids[0] is None:
                raise RuntimeError("rank 0 could not obtain an RCCL id")
            group = NativeGroup.from_rank(s, ids[0], rank, world)
            group.build_counts()
        except Exception as e:
            err = err or repr(e)
        if all_ranks_ok(err is None):
            exchange = {"kind": "native: mvhdp_group_sweep, RCCL all-reduce inside libmvhdp.so, %d row ranges pipelined with apply + tree rebuild" % group.info().exchange_chunks,
                        "rccl_version": int(group.info().rccl_version)}
        else:
            if group is not None:
                group.close()
            group = None
            exchange = {"kind": "fallback", "fallback_reason": err or "another rank failed"}
    if world > 1 and group is None:
        shard = GpuShard(s, device, host_staged=True)
        build_counts_all_reduce(shard)
        exchange["kind"] = ("fallback -> " if "fallback_reason" in exchange else "") + "torch.distributed (gloo) all-reduce on host copies, then apply"
    if world == 1:
        s.build_counts()
    setup_s = time.time() - t_setup

    phases = {}

    def step(idx, flags):
        t0 = time.perf_counter()
        if group is not None:
            st = group.sweep(idx, args.seed, flags)[0]
        elif shard is not None:
            st = sweep_all_reduce(shard, idx, args.seed, flags=flags, pipeline=False)
        else:
            st = s.sweep(idx, args.seed, flags=flags)
        phases["step_call_host"] = phases.get("step_call_host", 0.0) + (time.perf_counter() - t0) * 1e3
        phases["sweep_kernel"] = phases.get("sweep_kernel", 0.0) + st.sweep_kernel_ms
        phases["sweep_device_total"] = phases.get("sweep_device_total", 0.0) + st.total_ms
        if group is not None:
            phases["exchange_device"] = phases.get("exchange_device", 0.0) + group.info().last_exchange_ms
        phases["n"] = phases.get("n", 0) + 1
        return st

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # warm-up; a failure of the native exchange here is agreed on by all ranks, the model recounted, and the run continues on the
    # fallback path with the NEXT sweep index (nothing is re-sampled on top of a half-exchanged state)
    w = 0
    while w < args.warmup:
        err = None
        try:
            step(w, sweep_flags)
        except Exception as e:
            err = repr(e)
        w += 1
        if not all_ranks_ok(err is None):
            if group is None:
                raise SystemExit("sweep failed in the warm-up: %s" % (err or "on another rank"))
            group.close(); group = None
            shard = GpuShard(s, device, host_staged=True)
            build_counts_all_reduce(shard)            # recount from z on every rank + sum: a consistent model again
            exchange = {"kind": "fallback -> torch.distributed (gloo) all-reduce on host copies, then apply", "fallback_reason": err or "another rank failed"}
    phases.clear()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    last = None
    if args.batch and world == 1:
        sts = s.sweep_many(args.warmup, args.steps, args.seed, flags=sweep_flags)
        kernel_ms = [st.sweep_kernel_ms for st in sts]
        last = sts[-1]
        phases.update(step_call_host=(time.perf_counter() - t0) * 1e3, sweep_kernel=sum(kernel_ms), sweep_device_total=sum(st.total_ms for st in sts), n=args.steps)
    else:
        for k in range(args.steps):
            last = step(args.warmup + k, sweep_flags)
            kernel_ms.append(last.sweep_kernel_ms)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    value = total_tokens * args.steps / dt
    # roofline of the dominant kernel (sweep_fast_kernel<R>) on this rank: algorithmic bytes per launch /
    # its average duration, measured with hipEvents on the library's stream
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    bpt = algorithmic_bytes_per_token(K)
    achieved = local_tokens * bpt / avg_kernel_s / 1e9
    # the ceilings that can bind (fractions below 1) and, from the same committed PMC passes, `traffic`: the fabric bytes per second
    physical, issue = physical_rooflines(args.workload, local_tokens, avg_kernel_s, "live" if args.live else "deferred")
    traffic = physical["achieved"] if physical else None
    traffic_note = None
    if physical:
        traffic_note = (f"{physical['bytes_per_token']:.0f} B/token = TCC_EA0_RDREQ x 128 B (= 2 x FETCH_SIZE: the gfx950 correction, calibrated on "
                        "this access pattern in profiles/r02_fetch_calibration.txt) + WRITE_SIZE, Infinity-Cache hits included; PMC passes of "
                        "profiles/profile_r05.sh over the timed sweeps of this very command (a rocprofv3 run of its own, same build), the rate is this run's")
    binding = None
    if physical and issue:
        fabric = physical.get("frac_of_gather_ceiling") or physical["frac"]
        binding = "instruction issue (vector ALU)" if issue["frac"] >= fabric else "the fabric (gather of the count rows)"
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
                     "nominal_note": "achieved / frac are the fixed-byte yardstick of SURVEY 8d (4K+8 bytes per token whatever the kernel reads: "
                                     "it assumes 4-byte counts, the kernels gather 2-byte cells, so frac can pass 1); `physical` and `issue` are the "
                                     "ceilings that can bind, both below 1",
                     "physical": physical, "issue": issue, "binds": binding,
                     "avg_kernel_ms": avg_kernel_s * 1e3},
        "sweep": {"changed_frac": last.changed / max(1, last.tokens),
                  "branch_frac": {"new": last.new_mass_cnt / max(1, last.tokens),
                                  "doc": last.topic_doc_mass_cnt / max(1, last.tokens),
                                  "tree": last.word_ftree_mass_cnt / max(1, last.tokens)},
                  "exact_fallbacks": last.exact_fallbacks, "total_ms_last": last.total_ms},
        "setup_s": setup_s,
        # rank 0's milliseconds per step by phase: host wall time of the step call (plan, launches, one or two synchronisations),
        # device time of the sweep kernels, device time of the whole sweep, and (several GPUs) of the exchange: collectives + the
        # update and tree rebuild of the row ranges (mvhdp_group_info.last_exchange_ms)
        "phase_ms": {k: v / max(1, phases.get("n", 1)) for k, v in phases.items() if k != "n"},
        "exchange": exchange,
        "update_mode": ("live (live-rows form), %d segment(s) per sweep" % (args.live_segments or 1)) if args.live else "deferred (snapshot sweep, bit-reproducible)",
        "step_calls": "one mvhdp_sweep_many call for the K steps" if (args.batch and world == 1) else "one call per step",
        # What a sweep of each update mode is worth IN EVERY VIEW, in sweeps of the CPU restatement of the reference's thread topology: GPU
        # sweeps needed to reach the log-likelihood the reference reaches in one, from the curves of the configuration that is timed -- the
        # 200k-entity slice of C4 (K = 400: where live sweeps run on the 16-bit mirror) for K >= 256, C3 (K = 200) below, the 100k-entity slice
        # of C5 for C5 (a truncated HDP: profiles/r05_ll_curves_c5.md); two CPU chains, two seeds per live form (profiles/r05_ll_curves.md).  Tokens/s of different modes are comparable only after dividing by it.
        "reference_sweep_equivalent": sweep_equivalents(K, args.workload),
    }
    # order-independent fingerprint of the final global counts: must not depend on the number of shards
    nk_fp = [int(np.asarray(s.get_counts(m)[1], dtype=np.int64).dot(np.arange(1, K + 1, dtype=np.int64))) for m in range(M)]
    out["final_nk_fingerprint"] = nk_fp
    eq = out["reference_sweep_equivalent"]["gpu_sweeps_per_reference_sweep"]
    main_mode = "live" if args.live else "deferred"
    out["value_in_reference_sweeps"] = {main_mode: [value / eq[main_mode][1], value / eq[main_mode][0]]} if eq.get(main_mode) else {}
    if args.live_steps is None:
        args.live_steps = 10 if world == 1 else 0
    sec_warm = 3                                    # warm-up sweeps of a secondary mode (its kernels' flavours and thresholds settle)
    base_idx = args.warmup + args.steps

    def time_mode(flags, first_idx):
        for k in range(sec_warm):
            step(first_idx + k, flags)
        barrier()
        t1 = time.perf_counter()
        km = []
        mode_births[0] = 0
        for k in range(args.live_steps):
            st_ = step(first_idx + sec_warm + k, flags)
            km.append(st_.sweep_kernel_ms)
            mode_births[0] += int(st_.activations)
        barrier()
        mode_kernel_ms[0] = float(np.mean(km)) if km else 0.0
        return max_over_ranks(time.perf_counter() - t1)

    mode_kernel_ms = [0.0]
    mode_births = [0]                               # topics activated (UPD:263-270) during a mode's timed sweeps (a truncated HDP: C5)

    def mode_rooflines(mode):
        """the two ceilings of a secondary mode's sweep kernels, from the PMC / SQ passes over ITS window of this command (profiles/profile_r05.sh)"""
        if mode_kernel_ms[0] <= 0:
            return {}
        ph, iss = physical_rooflines(args.workload, local_tokens, mode_kernel_ms[0] / 1e3, mode)
        return {"roofline": {"avg_kernel_ms": mode_kernel_ms[0], "physical": ph, "issue": iss}} if ph else {"roofline": {"avg_kernel_ms": mode_kernel_ms[0], "physical": None}}

    if not args.live and args.live_steps > 0:
        # The other update mode, timed after (and outside) the K steps above: MVHDP_SWEEP_LIVE, the reference's own discipline
        # (atomics on the shared n_wk while the sweep samples, UPD:197-207; on the 16-bit mirror of the light rows where K >= 256)
        lf = SWEEP_LIVE | SWEEP_LIVE_SEGMENTS(args.live_segments)
        try:
            dl = time_mode(lf, base_idx)
            v = total_tokens * args.live_steps / dl
            out["live"] = {"value": v, "unit": "tokens/s", "steps": args.live_steps, "warmup": sec_warm,
                           "ms_per_step": dl / args.live_steps * 1e3,
                           "segments_per_sweep": args.live_segments or 1,
                           "topics_born_in_the_timed_sweeps": mode_births[0],
                           "note": "MVHDP_SWEEP_LIVE in its live-rows form (the tree branch samples from the word's live count row, on the 16-bit "
                                   "mirror where K >= 256; one segment), timed after the K deferred steps; not bit-reproducible; what a sweep of it "
                                   "is worth in every view: reference_sweep_equivalent"}
            out["live"].update(mode_rooflines("live"))
            if eq.get("live"):
                out["value_in_reference_sweeps"]["live"] = [v / eq["live"][1], v / eq["live"][0]]
        except Exception as e:                      # the secondary measurement must never cost the primary one
            out["live"] = {"error": repr(e)}
    if world == 1 and not args.live and args.live_steps > 0:
        # the deterministic middle ground, single handle only: a deferred sweep in 8 interleaved segments with the deltas applied
        # (and the trees rebuilt) in between (MVHDP_SWEEP_SEGMENT_APPLY): bit-exact against the oracle like the plain deferred sweep
        try:
            sf = SWEEP_SEGMENT_APPLY | SWEEP_LIVE_SEGMENTS(8)
            dsg = time_mode(sf, base_idx + sec_warm + args.live_steps)
            v = total_tokens * args.live_steps / dsg
            out["segmented"] = {"value": v, "unit": "tokens/s", "steps": args.live_steps, "warmup": sec_warm,
                                "ms_per_step": dsg / args.live_steps * 1e3, "segments": 8,
                                "note": "MVHDP_SWEEP_SEGMENT_APPLY: deferred sweep in 8 segments, deltas applied and trees rebuilt in "
                                        "between; deterministic (oracle-checked); what a sweep of it is worth in every view: reference_sweep_equivalent"}
            out["segmented"].update(mode_rooflines("segmented"))
            if eq.get("segmented"):
                out["value_in_reference_sweeps"]["segmented"] = [v / eq["segmented"][1], v / eq["segmented"][0]]
        except Exception as e:
            out["segmented"] = {"error": repr(e)}
        # ... and the deferred sweep once more, at the chain age the two modes above were timed at: `value` is taken over sweeps
        # W .. W+K-1 from a random start (the driver's window: the slow first dozen sweeps are in it), the modes on an older chain --
        # a like-for-like ratio needs this line
        try:
            dd = time_mode(0, base_idx + 2 * (sec_warm + args.live_steps))
            v = total_tokens * args.live_steps / dd
            out["deferred_same_age"] = {"value": v, "unit": "tokens/s", "steps": args.live_steps, "warmup": sec_warm, "ms_per_step": dd / args.live_steps * 1e3, **mode_rooflines("deferred_same_age"),
                                        "note": "the deferred mode timed like `live` and `segmented`, after them: the denominator for comparing the modes"}
            for k in ("live", "segmented"):
                if "value" in out.get(k, {}):
                    out[k]["vs_deferred_same_age"] = out[k]["value"] / v
        except Exception as e:
            out["deferred_same_age"] = {"error": repr(e)}
    if group is not None:
        group.close()
    if shard is not None:
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
