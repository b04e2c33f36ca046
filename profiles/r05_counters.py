"""Per-token counters of the sweep kernels over a window of sweeps, from one rocprofv3 --pmc run (profiles/profile_r05.sh).

A sweep starts at its MARKER kernel: draw_p_kernel (the view weights, drawn once at the head of every sweep of a multi-view model, whatever
the update mode) or, for a single-view model (C2), build_trees_kernel (once per deferred sweep).  The counters of every sweep kernel (all
classes) of sweeps [first, first + n) are summed and divided by n x tokens; the kernel time of the same window comes from the kernel
trace of the same run."""
import collections
import csv
import glob
import os


def _is_sweep(name):
    return "sweep_fast_kernel" in name or "sweep_kernel" in name


def window(out_dir, tokens, first, n, marker="draw_p_kernel"):
    files = sorted(glob.glob(f"{out_dir}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows, names = collections.defaultdict(dict), {}
    for r in csv.DictReader(open(files[-1])):
        k = int(r["Dispatch_Id"])
        names[k] = r["Kernel_Name"]
        if _is_sweep(r["Kernel_Name"]):
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tot, sweep = collections.defaultdict(float), -1
    for k in sorted(names):
        if marker in names[k]:
            sweep += 1
        if k in rows and first <= sweep < first + n:
            for c, v in rows[k].items():
                tot[c] += v
    per = {c: v / (n * tokens) for c, v in tot.items()}
    per["_sweeps_seen"] = sweep + 1
    kt = sorted(glob.glob(f"{out_dir}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if kt:
        ms, span0, span1, sweep = 0.0, None, None, -1
        for r in sorted(csv.DictReader(open(kt[-1])), key=lambda r: int(r["Start_Timestamp"])):
            if marker in r["Kernel_Name"]:
                sweep += 1
            if _is_sweep(r["Kernel_Name"]) and first <= sweep < first + n:
                ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                span0 = int(r["Start_Timestamp"]) if span0 is None else span0
                span1 = max(span1 or 0, int(r["End_Timestamp"]))
        per["_kernel_ms_sum_per_sweep"] = ms / n                     # (class kernels that run side by side are counted each: a sum, not a span)
        per["_kernel_span_ms_per_sweep"] = ((span1 - span0) / 1e6 / n) if span0 is not None else None
    return per
