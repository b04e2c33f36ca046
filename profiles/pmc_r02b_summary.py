"""Per-token memory-side counters of the sweep kernels over the timed sweeps of `bench.py --steps 20 --warmup 5` (profiles/pmc_r02b.sh).
A sweep is one primary kernel (sweep_fast_kernel<1|2|4,...>) plus its overflow passes; the first 5 sweeps are the warm-up."""
import collections
import csv
import glob
import json
import re
import sys

out_dir = sys.argv[1]
TOK = 147225025
WARM, STEPS = 5, 20
tot = collections.defaultdict(float)
for d in sorted(glob.glob(f"{out_dir}/s[0-9]*")):
    if d.endswith(".log"):
        continue
    rows = collections.defaultdict(dict)
    names = {}
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "sweep_fast_kernel" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]:
                k = int(r["Dispatch_Id"])
                rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[k] = r["Kernel_Name"]
    sweep = -1
    for k in sorted(rows):
        m = re.search(r"sweep_fast_kernel<(\d+)", names[k])
        if m and int(m.group(1)) <= 4:
            sweep += 1                      # a primary kernel opens a sweep
        if WARM <= sweep < WARM + STEPS:
            for c, v in rows[k].items():
                tot[c] += v
    tot["_sweeps_seen_" + d.rsplit("/", 1)[1]] = sweep + 1
per = {c: v / (STEPS * TOK) for c, v in tot.items() if not c.startswith("_")}
res = {
    "source": "profiles/pmc_r02b.sh: separate rocprofv3 --kernel-trace --pmc passes of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --live-steps 0` "
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
json.dump(res, open(f"{out_dir}/r02b_c4_pmc_summary.json", "w"), indent=1)
