"""Reads gpurun_out/calib_<tag>/ (profiles/calib.sh) and prints/saves the calibration of the memory-side read
counters: for each microbenchmark case the bytes touched at 64- and 128-byte granularity (known by construction)
next to what the counters report, then the same counters for the C4 sweep kernel with the calibrated reading."""
import collections
import csv
import glob
import json
import sys

out_dir = sys.argv[1]
TOK = 147225025


def counters(d, name_filter):
    """{counter: [value per dispatch, in dispatch order]} for kernels whose name passes the filter."""
    res = collections.defaultdict(lambda: collections.OrderedDict())
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if name_filter(r["Kernel_Name"]):
                key = int(r["Dispatch_Id"])
                res[r["Counter_Name"]][key] = res[r["Counter_Name"]].get(key, 0.0) + float(r["Counter_Value"])
    return {c: [v for _, v in sorted(dd.items())] for c, dd in res.items()}


truth = [json.loads(l) for l in open(f"{out_dir}/truth.jsonl") if l.startswith("{")]
micro = {}
for d in sorted(glob.glob(f"{out_dir}/m[0-9]*")):
    if d.endswith(".log"):
        continue
    micro.update(counters(d, lambda n: n.startswith("calib_") or "calib_" in n))
summary = {"microbench": [], "sweep": {}}
print("case                          | true 64B-units KB | true 128B-units KB | FETCH_SIZE KB | RDREQ | 32B | 64B | 128B | BUBBLE | L2 hit | L2 miss | TCP->TCC")
for i, t in enumerate(truth):
    row = {"case": t["case"], "served_from": t["served_from"], "ms": t["ms"], "useful_KB": t["useful_bytes"] / 1024,
           "true_units64_KB": t["units64"] * 64 / 1024, "true_units128_KB": t["units128"] * 128 / 1024,
           "units64": t["units64"], "units128": t["units128"]}
    for c, vals in micro.items():
        if i < len(vals):
            row[c] = vals[i]
    if "FETCH_SIZE" in row:
        row["FETCH_SIZE_over_true64"] = row["FETCH_SIZE"] / row["true_units64_KB"]
        row["FETCH_SIZE_over_true128"] = row["FETCH_SIZE"] / row["true_units128_KB"]
    if "TCC_EA0_RDREQ" in row:
        row["RDREQ_over_units64"] = row["TCC_EA0_RDREQ"] / t["units64"]
        row["RDREQ_over_units128"] = row["TCC_EA0_RDREQ"] / t["units128"]
    summary["microbench"].append(row)
    g = lambda k: ("%.3g" % row[k]) if k in row else "-"
    print(f"{t['case']:30s}| {row['true_units64_KB']:.4g} | {row['true_units128_KB']:.4g} | {g('FETCH_SIZE')} | {g('TCC_EA0_RDREQ')} | "
          f"{g('TCC_EA0_RDREQ_32B')} | {g('TCC_EA0_RDREQ_64B')} | {g('TCC_EA0_RDREQ_128B')} | {g('TCC_BUBBLE')} | {g('TCC_HIT')} | {g('TCC_MISS')} | {g('TCP_TCC_READ_REQ')}"
          f"   FETCH/true64={g('FETCH_SIZE_over_true64')} FETCH/true128={g('FETCH_SIZE_over_true128')} RDREQ/u64={g('RDREQ_over_units64')} RDREQ/u128={g('RDREQ_over_units128')}")

sw = {}
for d in sorted(glob.glob(f"{out_dir}/s[0-9]*")):
    if d.endswith(".log"):
        continue
    # the dominant sweep kernel: the variant with the largest number of dispatches among sweep_fast_kernel<...>
    allc = counters(d, lambda n: "sweep_fast_kernel<2" in n)
    for c, vals in allc.items():
        vals = vals[2:] if len(vals) > 2 else vals          # drop the warm-up sweeps
        sw[c] = sum(vals) / len(vals)
print("\nC4 sweep kernel (sweep_fast_kernel<2>), mean of the timed launches, per token:")
for c in sorted(sw):
    print(f"  {c:28s} {sw[c]:16.0f}   per token {sw[c] / TOK:10.3f}")
summary["sweep"] = {"per_launch": sw, "per_token": {c: v / TOK for c, v in sw.items()}, "tokens_per_launch": TOK}
json.dump(summary, open(f"{out_dir}/summary.json", "w"), indent=1)
