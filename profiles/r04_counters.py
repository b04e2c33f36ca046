"""Per-token counters of the sweep kernels over a window of sweeps, from one rocprofv3 --pmc run (profiles/profile_r04.sh).

  python3 profiles/r04_counters.py DIR TOKENS_PER_SWEEP FIRST_SWEEP N_SWEEPS [kernel-name filter]

A sweep is what runs between two build_trees_kernel launches of a DEFERRED chain (one rebuild per sweep): the counters of every
sweep kernel (all classes) of sweeps [FIRST, FIRST + N) are summed and divided by N x TOKENS.  Prints one JSON object."""
import collections
import csv
import glob
import json
import os
import sys


def window(out_dir, tokens, first, n, name_filter=None):
    files = sorted(glob.glob(f"{out_dir}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows, names = collections.defaultdict(dict), {}
    for r in csv.DictReader(open(files[-1])):
        k = int(r["Dispatch_Id"])
        names[k] = r["Kernel_Name"]
        if ("sweep_fast_kernel" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]) and (not name_filter or name_filter in r["Kernel_Name"]):
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tot, sweep = collections.defaultdict(float), -1
    for k in sorted(names):
        if names[k].startswith("build_trees_kernel"):
            sweep += 1
        if k in rows and first <= sweep < first + n:
            for c, v in rows[k].items():
                tot[c] += v
    per = {c: v / (n * tokens) for c, v in tot.items()}
    per["_sweeps_seen"] = sweep + 1
    # kernel time of the same window from the kernel trace of the same run
    kt = sorted(glob.glob(f"{out_dir}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if kt:
        ms, sweep = 0.0, -1
        for r in sorted(csv.DictReader(open(kt[-1])), key=lambda r: int(r["Start_Timestamp"])):
            if r["Kernel_Name"].startswith("build_trees_kernel"):
                sweep += 1
            if ("sweep_fast_kernel" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]) and first <= sweep < first + n:
                ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per["_kernel_ms_sum_per_sweep"] = ms / n
    return per


if __name__ == "__main__":
    d, tokens, first, n = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    print(json.dumps(window(d, tokens, first, n, sys.argv[5] if len(sys.argv) > 5 else None)))
