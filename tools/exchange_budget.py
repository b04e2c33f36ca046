#!/usr/bin/env python3
"""What the exchange of an N-GPU run puts on the wire per sweep, and what packing it costs, measured on ONE GPU through the library's own
one-process-per-GPU path (mvhdp_group_create_rank): R ranks (default 2) of an N-way document cut (default 8) of a workload, every rank a
process on cuda:0, the collective being tests/native/fake_rccl.c (MVHDP_RCCL_LIB; the real RCCL refuses two ranks on one device).  The
wire itself cannot be timed here; the bytes a rank hands to the collective (mvhdp_group_info.last_exchange_bytes) and the kernels either
side of it (pack_rows_kernel / unpack_rows_kernel, rank 0 under rocprofv3 --kernel-trace --stats) can.

  python tools/exchange_budget.py run --workload C4 --of 8 --ranks 2 --steps 8 --warmup 4 --out gpurun_out/exchange_c4.json

The counts every rank starts from are those of the R shards that run (not of all N): the delta rows, their packing and the bytes are
what they are at N ranks -- the table's shape does not depend on the cut -- the chain is not C4's (nothing here is a throughput figure).
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    from mvtopicmodel_amd import NativeGroup, NativeSampler, synth
    from mvtopicmodel_amd.java_init import init_assignments
    from mvtopicmodel_amd.native import Hyper
    cfg = synth.CONFIGS[args.workload]
    K, V = cfg["K"], cfg["V"]
    M = len(V)
    c = synth.make_config(args.workload)
    inactive, K_init = synth.config_inactive(args.workload)
    z0 = init_assignments(K_init, c.doc_off, seed=1)
    tot = sum(np.diff(c.doc_off[m]) for m in range(M))
    lo, hi = synth.shard_bounds(tot, args.of)[args.rank]
    sub = c.slice_docs(lo, hi)
    s = NativeSampler(K, V, device=0, doc_id_base=lo)
    for m in range(M):
        s.set_corpus(m, sub.doc_off[m], sub.tokens[m]); s.set_assignments(m, z0[m][c.doc_off[m][lo]:c.doc_off[m][hi]])
    s.set_hyper(Hyper.defaults(K, V, inactive=inactive)); s.build_counts()
    uid_path = os.path.join(args.workdir, "uid")
    if args.rank == 0:
        uid = NativeGroup.unique_id()
        open(uid_path + ".tmp", "wb").write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 600:
                raise SystemExit("no id from rank 0")
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    g = NativeGroup.from_rank(s, uid, args.rank, args.ranks)
    g.build_counts()
    rows = []
    for it in range(args.warmup + args.steps):
        t0 = time.perf_counter()
        st = g.sweep(it, 1, 0)[0]
        t1 = time.perf_counter()
        gi = g.info()
        rows.append({"sweep": it, "packed": int(gi.exchange_packed), "bytes": int(gi.last_exchange_bytes), "exchange_ms_with_the_fake_collective": float(gi.last_exchange_ms),
                     "sweep_kernel_ms": float(st.sweep_kernel_ms), "call_ms": (t1 - t0) * 1e3, "changed": int(st.changed)})
        print(f"rank {args.rank} sweep {it}: packed {rows[-1]['packed']} bytes {rows[-1]['bytes']}", flush=True)
    json.dump({"rank": args.rank, "tokens": int(sum(int(sub.doc_off[m][-1]) for m in range(M))), "sweeps": rows,
               "table_bytes_int32": int((sum(V) + M) * K * 4)}, open(os.path.join(args.workdir, f"rank{args.rank}.json"), "w"))
    g.close(); s.close()


def run(args):
    tmp = tempfile.mkdtemp(prefix="xb_")
    fake = os.path.join(tmp, "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "fake_rccl.c"), "-o", fake,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    res = {"workload": args.workload, "of": args.of, "ranks_run": args.ranks, "steps": args.steps, "warmup": args.warmup, "forms": {}}
    for x16 in (0, 1):
        wd = os.path.join(tmp, f"x{x16}")
        os.makedirs(wd)
        env = dict(os.environ, MVHDP_RCCL_LIB=fake, MVHDP_EXCHANGE16=str(x16), FAKE_RCCL_TIMEOUT_MS="120000", FAKE_RCCL_SLOT_BYTES=str(64 << 20), TMPDIR="/tmp")
        procs = []
        for r in range(args.ranks):
            cmd = [sys.executable, os.path.abspath(__file__), "worker", "--workload", args.workload, "--of", str(args.of), "--ranks", str(args.ranks),
                   "--rank", str(r), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workdir", wd]
            if r == 0 and not args.no_profile:
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(wd, "prof"), "--"] + cmd
            procs.append(subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        try:
            outs = [p.communicate(timeout=args.timeout)[0] for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if any(p.returncode for p in procs):
            print("\n".join(o[-2000:] for o in outs)); raise SystemExit("a rank failed")
        r0 = json.load(open(os.path.join(wd, "rank0.json")))
        timed = r0["sweeps"][args.warmup:]
        form = {"exchange_packed_flags": [w["packed"] for w in r0["sweeps"]], "bytes_on_the_wire_per_sweep": float(np.mean([w["bytes"] for w in timed])),
                "table_bytes_int32": r0["table_bytes_int32"], "rank0_tokens": r0["tokens"],
                "rank0_sweep_kernel_ms": float(np.mean([w["sweep_kernel_ms"] for w in timed]))}
        ks = sorted(glob.glob(os.path.join(wd, "prof", "**", "*kernel_stats.csv"), recursive=True))
        if ks:
            per = {}
            allk = {}
            for row in csv.DictReader(open(ks[-1])):
                allk[row["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80]] = round(float(row["TotalDurationNs"]) / 1e6 / (args.warmup + args.steps), 4)
            form["all_kernels_rank0_ms_per_sweep"] = dict(sorted(allk.items(), key=lambda kv: -kv[1])[:30])
            for row in csv.DictReader(open(ks[-1])):
                n = row["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if any(t in n for t in ("pack_rows", "unpack_rows", "add_remote", "add_into", "apply_delta", "apply_nk")):
                    per[n] = {"calls": int(row["Calls"]), "ms_per_sweep_all_sweeps": float(row["TotalDurationNs"]) / 1e6 / (args.warmup + args.steps)}
            form["exchange_side_kernels_rank0"] = per
        res["forms"]["packed (two 16-bit deltas per word for the light rows)" if x16 else "int32"] = form
    a, b = res["forms"]["int32"], res["forms"]["packed (two 16-bit deltas per word for the light rows)"]
    res["bytes_ratio_packed_to_int32"] = b["bytes_on_the_wire_per_sweep"] / a["bytes_on_the_wire_per_sweep"]
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["run", "worker"])
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--of", type=int, default=8)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--out", default="gpurun_out/exchange_budget.json")
    ap.add_argument("--timeout", type=int, default=500)
    ap.add_argument("--no-profile", action="store_true")
    a = ap.parse_args()
    {"run": run, "worker": worker}[a.cmd](a)
