#!/usr/bin/env python3
"""Timeline of the kernels of the last sweeps of a run, from a `rocprofv3 --kernel-trace` CSV: start offset, duration, queue, and the
gaps between consecutive kernels -- where a segment border or a short launch loses its time.

  rocprofv3 --kernel-trace -d gpurun_out/prof -o trace -- python3 tools/mode_times.py --mode seg8 --sweeps 30
  python tools/kernel_timeline.py gpurun_out/prof --anchor draw_p_kernel --last 2
"""
import argparse
import csv
import glob
import json
import os
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--anchor", default="draw_p_kernel", help="kernel that starts a sweep call")
    ap.add_argument("--last", type=int, default=1, help="how many of the last sweeps to print")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *kernel_trace.csv under " + a.dir
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                         r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if r[2].startswith(a.anchor)]
    assert anchors, "anchor kernel not found"
    out = []
    for k in range(max(0, len(anchors) - a.last), len(anchors)):
        lo = anchors[k]
        hi = anchors[k + 1] if k + 1 < len(anchors) else len(rows)
        t0 = rows[lo][0]
        sweep = []
        busy_end = t0
        for s, e, n, q, v, g in rows[lo:hi]:
            sweep.append({"kernel": n, "queue": q, "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3, "gap_before_us": (s - busy_end) / 1e3, "vgpr": v, "grid": g})
            busy_end = max(busy_end, e)
        out.append({"sweep": k, "span_us": (busy_end - t0) / 1e3, "kernels": sweep})
    if a.json:
        print(json.dumps(out))
        return
    for sw in out:
        print("sweep call %d: %.1f us" % (sw["sweep"], sw["span_us"]))
        for kr in sw["kernels"]:
            print("  %9.1f  +%8.1f us  gap %7.1f  q%-3s vgpr %-4s grid %-8s %s" % (kr["start_us"], kr["dur_us"], kr["gap_before_us"], kr["queue"], kr["vgpr"], kr["grid"], kr["kernel"]))


if __name__ == "__main__":
    main()
