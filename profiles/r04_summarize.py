"""Turns the rocprofv3 runs of profiles/profile_r04.sh into gpurun_out/r04_<w>_counters.json (per workload) and, with `merge`, into
gpurun_out/r04_roofline_inputs.json -- what bench.py's roofline.physical / roofline.issue objects are computed from."""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from r04_counters import window

OUT = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(W, tokens, warm, steps):
    w = W.lower()
    res = {"workload": W, "tokens_per_sweep": tokens}

    def block(tag, first, n, suffix=""):
        rd = window(f"{OUT}/{w}{suffix}_rd", tokens, first, n)
        wr = window(f"{OUT}/{w}{suffix}_wr", tokens, first, n)
        sq = window(f"{OUT}/{w}{suffix}_sq", tokens, first, n)
        return {"window": f"sweeps {first}..{first + n - 1}", "fabric_read_requests_per_token": rd.get("TCC_EA0_RDREQ"),
                "fabric_read_bytes_per_token": rd.get("TCC_EA0_RDREQ", 0.0) * 128.0,
                "write_bytes_per_token": wr.get("WRITE_SIZE", 0.0) * 1024.0,
                "valu_busy_cycles_per_token": sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0,
                "scalar_busy_cycles_per_token": sq.get("SQ_ACTIVE_INST_SCA", 0.0) * 4.0,
                "valu_insts_per_token": sq.get("SQ_INSTS_VALU"), "scalar_insts_per_token": sq.get("SQ_INSTS_SALU"),
                "wave_cycles_per_token": sq.get("SQ_WAVE_CYCLES", 0.0) * 4.0,
                "kernel_ms_per_sweep_profiled": {"rdreq_pass": rd.get("_kernel_ms_sum_per_sweep"), "sq_pass": sq.get("_kernel_ms_sum_per_sweep")},
                "sweeps_seen": {"rd": rd.get("_sweeps_seen"), "wr": wr.get("_sweeps_seen"), "sq": sq.get("_sweeps_seen")},
                "pmc_source": f"profiles/profile_r04.sh: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ / WRITE_SIZE passes ({tag}); a request = 128 B (profiles/r02_fetch_calibration.txt)",
                "sq_source": f"profiles/profile_r04.sh: rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA ... pass ({tag}); the counters tick once per four cycles"}

    res["deferred"] = block(f"`bench.py --workload {W} --steps {steps} --warmup {warm} --no-cpu-baseline --live-steps 0`, the timed sweeps", warm, steps)
    if os.path.isdir(f"{OUT}/{w}_set_rd"):
        res["settled"] = block("`tools/per_sweep_times.py --workload C4 --sweeps 50`, sweeps 40-49", 40, 10, "_set")
    if os.path.isdir(f"{OUT}/{w}_fs"):
        fs = window(f"{OUT}/{w}_fs", tokens, warm, steps)
        hm = window(f"{OUT}/{w}_hit", tokens, warm, steps)
        res["deferred"]["FETCH_SIZE_bytes_per_token_raw"] = fs.get("FETCH_SIZE", 0.0) * 1024.0
        res["deferred"]["TCC_HIT_per_token"] = hm.get("TCC_HIT")
        res["deferred"]["TCC_MISS_per_token"] = hm.get("TCC_MISS")
    json.dump(res, open(f"{ROOT}/gpurun_out/r04_{w}_counters.json", "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if "source" not in kk}) for k, v in res.items()}, indent=1))


def merge():
    out = {"workloads": {}}
    for f in sorted(glob.glob(f"{ROOT}/gpurun_out/r04_c?_counters.json")):
        j = json.load(open(f))
        out["workloads"][j["workload"]] = {k: v for k, v in j.items() if k in ("deferred", "settled")}
    try:
        txt = open(f"{ROOT}/gpurun_out/r04_row_gather_ceiling.txt").read()
        m = re.search(r"sparse 2-byte gather:.*= +([0-9.]+) GB/s", txt)
        out["gather_ceiling_GBs"] = float(m.group(1))
        out["gather_ceiling_source"] = ("tools/microbench/row_gather_ceiling.hip (profiles/r04_row_gather_ceiling.txt): 7 waves per SIMD doing nothing but the sweep "
                                        "kernel's gather -- 45 sorted 2-byte cells of random 800-byte rows of a 48 MB table -- lines touched x 128 B / s")
    except Exception as e:
        out["gather_ceiling_GBs"] = None
        out["gather_ceiling_source"] = f"not measured ({e!r})"
    json.dump(out, open(f"{ROOT}/gpurun_out/r04_roofline_inputs.json", "w"), indent=1)


if sys.argv[2] == "merge":
    merge()
else:
    one(sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
