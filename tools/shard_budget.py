#!/usr/bin/env python3
"""What one rank of an N-GPU run spends per sweep, measured on ONE GPU: rank 0's document shard of the workload
(same cut as bench.py), the same call sequence as mvtopicmodel_amd.dist.sweep_all_reduce (sweep with NO_APPLY ->
[all-reduce, not run here] -> apply_delta), host wall time and device time of every phase.  The collective itself cannot
be timed on one GPU; everything else of the per-rank budget can (is >= 6x at 8 GPUs reachable?).

  python tools/shard_budget.py --workload C4 --of 8 --steps 20 --warmup 5
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def exact(args):
    import torch
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.dist import PIPELINE_CHUNKS, GpuShard
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_REUSE_TREES
    cfg = synth.CONFIGS[args.workload]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(args.workload)
    inactive, K_init = synth.config_inactive(args.workload)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    tot = sum(np.diff(c.doc_off[m]) for m in range(M))
    bounds = synth.shard_bounds(tot, args.of)
    shards = []
    for lo, hi in bounds:
        sub = c.slice_docs(lo, hi)
        s = NativeSampler(K, V, doc_id_base=lo)
        for m in range(M):
            s.set_corpus(m, sub.doc_off[m], sub.tokens[m]); s.set_assignments(m, z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
        s.set_hyper(Hyper.defaults(K, V, inactive=inactive)); s.build_counts()
        shards.append(GpuShard(s, "cuda:0"))
    torch.cuda.synchronize()
    total = shards[0].counts.clone()
    for g in shards[1:]:
        total += g.counts
    for g in shards:
        g.counts.copy_(total)
    torch.cuda.synchronize()
    for g in shards:
        g.counts_written()
    chunks, _ = shards[0].row_chunks(PIPELINE_CHUNKS)
    ph = {"sweep_call_host": 0.0, "sweep_kernel_dev": 0.0, "apply_and_trees_host": 0.0}
    per_sweep = []                                    # rank 0's sweep-kernel ms, every sweep (warm-up included)
    for it in range(args.warmup + args.steps):
        sts = []
        for r, g in enumerate(shards):
            t0 = time.perf_counter()
            st = g.sweep_local(it, 1, SWEEP_REUSE_TREES if g.trees_current() else 0)
            t1 = time.perf_counter()
            sts.append(st)
            if r == 0:
                per_sweep.append(round(st.sweep_kernel_ms, 3))
            if r == 0 and it >= args.warmup:
                ph["sweep_call_host"] += (t1 - t0) * 1e3; ph["sweep_kernel_dev"] += st.sweep_kernel_ms
        torch.cuda.synchronize()
        total = shards[0].delta.clone()
        for g in shards[1:]:
            total += g.delta
        for g in shards:
            g.delta.copy_(total)
        torch.cuda.synchronize()
        # UPD:263-270 across the shards: the first activating delta in (entity, view, position) order wins on every replica
        from mvtopicmodel_amd.dist import decode_activation
        topic, view = decode_activation(min(int(st.activation_key) for st in sts))
        for r, g in enumerate(shards):
            t0 = time.perf_counter()
            g.s.apply_delta_begin()
            for r0, r1 in chunks:
                g.s.apply_delta_rows(r0, r1)
            g.s.apply_delta_end(topic, view)
            if r == 0 and it >= args.warmup:
                ph["apply_and_trees_host"] += (time.perf_counter() - t0) * 1e3
    out = {k: v / args.steps for k, v in ph.items()}
    nk_fp = [int(np.asarray(shards[0].s.get_counts(m)[1], dtype=np.int64).dot(np.arange(1, K + 1, dtype=np.int64))) for m in range(M)]
    out.update(rank0_sweep_kernel_ms_per_sweep=per_sweep, workload=args.workload, ranks=args.of, shard0_tokens=int(tot[bounds[0][0]:bounds[0][1]].sum()), final_nk_fingerprint=nk_fp,
               note="all shards on one GPU, true global counts; rank 0's calls timed; the collective itself is not (device-side sum instead)")
    print(json.dumps(out))
    for g in shards:
        g.close(); g.s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--of", type=int, default=8, help="number of ranks the corpus is cut for; rank 0's shard is run")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--live", action="store_true")
    ap.add_argument("--plain", action="store_true", help="the unpipelined sequence: sweep (rebuilds the trees) -> apply_delta")
    ap.add_argument("--exact", action="store_true",
                    help="hold ALL N shards on this one GPU and run the real N-shard computation (deltas summed on the device as the "
                         "all-reduce would): rank 0's calls are timed against true global counts, not against scaled local ones")
    args = ap.parse_args()
    import torch
    from mvtopicmodel_amd import NativeSampler, synth
    from mvtopicmodel_amd.dist import GpuShard
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper, SWEEP_LIVE

    if args.exact:
        return exact(args)
    cfg = synth.CONFIGS[args.workload]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    doc_tokens = synth.config_doc_token_counts(args.workload)
    lo, hi = synth.shard_bounds(doc_tokens, args.of)[0]
    c = synth.make_config(args.workload, doc_lo=lo, doc_hi=hi)
    inactive, K_init = synth.config_inactive(args.workload)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    s = NativeSampler(K, V, doc_id_base=lo)
    for m in range(M):
        s.set_corpus(m, c.doc_off[m], c.tokens[m]); s.set_assignments(m, z0[m])
    s.set_hyper(Hyper.defaults(K, V, inactive=inactive))
    s.build_counts()
    # global-sized counts (the replica every rank holds): scale the shard's counts so that n_k has the magnitude of the full corpus
    shard = GpuShard(s, "cuda:0")
    with shard.on_stream():
        shard.counts.mul_(args.of)
    torch.cuda.synchronize()
    s.counts_written()
    flags = SWEEP_LIVE if args.live else 0
    ph = {"sweep_call_host": 0.0, "sweep_kernel_dev": 0.0, "sweep_total_dev": 0.0, "apply_host": 0.0}
    from mvtopicmodel_amd.dist import PIPELINE_CHUNKS
    from mvtopicmodel_amd.native import SWEEP_REUSE_TREES
    chunks, _ = shard.row_chunks(PIPELINE_CHUNKS)
    for it in range(args.warmup + args.steps):
        t0 = time.perf_counter()
        if args.plain:
            st = shard.sweep_local(it, 1, flags)
            t1 = time.perf_counter()
            # (the all-reduce of shard.delta would run here, on the same stream)
            shard.apply(-1, -1)
        else:
            st = shard.sweep_local(it, 1, flags | (SWEEP_REUSE_TREES if shard.trees_current() else 0))
            t1 = time.perf_counter()
            # (the chunked all-reduce would be in flight here; each chunk's rows are applied and their trees rebuilt as it lands)
            s.apply_delta_begin()
            for r0, r1 in chunks:
                s.apply_delta_rows(r0, r1)
            s.apply_delta_end(-1, -1)
        t2 = time.perf_counter()
        if it >= args.warmup:
            ph["sweep_call_host"] += (t1 - t0) * 1e3; ph["apply_host"] += (t2 - t1) * 1e3
            ph["sweep_kernel_dev"] += st.sweep_kernel_ms; ph["sweep_total_dev"] += st.total_ms
    out = {k: v / args.steps for k, v in ph.items()}
    out.update(rank0_sweep_kernel_ms_per_sweep=per_sweep, workload=args.workload, ranks=args.of, shard_entities=c.D, shard_tokens=c.total_tokens,
               per_rank_ms_without_collective=out["sweep_call_host"] + out["apply_host"],
               delta_bytes=int(shard.delta.numel() * 4), mode="live" if args.live else "deferred",
               sequence="plain" if args.plain else "pipelined (apply + tree rebuild by row ranges, hidden behind the collective on a real run)")
    print(json.dumps(out))
    shard.close(); s.close()


if __name__ == "__main__":
    main()
