"""Per-token memory-side counters of the sweep kernels over the timed sweeps of `bench.py --steps 20 --warmup 5` (profiles/pmc_r02b.sh).
A sweep is what runs between two build_trees_kernel launches (all sweep kernels of every class / overflow pass); the first 5
sweeps are the warm-up."""
import collections
import csv
import glob
import json
import os
import re
import sys

out_dir = sys.argv[1]
TOK = 147225025
WARM, STEPS = int(os.environ.get("PMC_WARM", "5")), int(os.environ.get("PMC_STEPS", "20"))   # the bench command's --warmup / --steps
tot = collections.defaultdict(float)
for d in sorted(glob.glob(f"{out_dir}/s[0-9]*")):
    if d.endswith(".log"):
        continue
    rows = collections.defaultdict(dict)
    names = {}
    files = sorted(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:                     # (gpurun_out/ keeps the files of earlier runs of the recipe: the newest only)
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            names[k] = r["Kernel_Name"]
            if "sweep_fast_kernel" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]:
                rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # a deferred sweep starts with ONE build_trees_kernel launch (then the view weights, a classify pass if any, the sweep
    # kernels of every class / overflow pass, the apply): the sweep kernels between two of them belong to one sweep
    sweep = -1
    for k in sorted(names):
        if names[k].startswith("build_trees_kernel"):
            sweep += 1
        if k in rows and WARM <= sweep < WARM + STEPS:
            for c, v in rows[k].items():
                tot[c] += v
    tot["_sweeps_seen_" + d.rsplit("/", 1)[1]] = sweep + 1
per = {c: v / (STEPS * TOK) for c, v in tot.items() if not c.startswith("_")}
res = {
    "source": os.environ.get("PMC_SOURCE", "profiles/pmc_r02b.sh") + ": separate rocprofv3 --kernel-trace --pmc passes of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --live-steps 0` "
              "(C4, 1 GPU), all sweep kernels of the 20 timed sweeps; counter calibration in profiles/r02_fetch_calibration.txt",
    "tokens_per_launch": TOK,
    "per_token_raw": per,
    "sweeps_seen": {k[13:]: v for k, v in tot.items() if k.startswith("_")},
    "fabric_read_requests_per_token": per.get("TCC_EA0_RDREQ"),
    "fetch_bytes_per_token": per.get("TCC_EA0_RDREQ", 0.0) * 128.0,
    "fetch_bytes_per_token_raw_FETCH_SIZE": per.get("FETCH_SIZE", 0.0) * 1024.0,
    "write_bytes_per_token": per.get("WRITE_SIZE", 0.0) * 1024.0,
    "algorithmic_bytes_per_token": 1608,
}
res["measured_over_algorithmic"] = (res["fetch_bytes_per_token"] + res["write_bytes_per_token"]) / 1608.0
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out_dir}/{os.environ.get('PMC_TAG', 'r02b')}_c4_pmc_summary.json", "w"), indent=1)
