"""Turns the rocprofv3 runs of profiles/profile_r05.sh into gpurun_out/r05_roofline_inputs.json: per workload AND per update mode of the
bench line (deferred over the driver's window, live, segmented, the deferred sweep at their chain age), what bench.py's
roofline.physical / roofline.issue objects are computed from."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from r05_counters import window

OUT = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def block(w, tokens, tag, first, n, marker):
    rd = window(f"{OUT}/{w}_rd", tokens, first, n, marker)
    wr = window(f"{OUT}/{w}_wr", tokens, first, n, marker)
    sq = window(f"{OUT}/{w}_sq", tokens, first, n, marker)
    return {"window": f"sweeps {first}..{first + n - 1} of the command", "fabric_read_requests_per_token": rd.get("TCC_EA0_RDREQ"),
            "fabric_read_bytes_per_token": rd.get("TCC_EA0_RDREQ", 0.0) * 128.0,
            "write_bytes_per_token": wr.get("WRITE_SIZE", 0.0) * 1024.0,
            "valu_busy_cycles_per_token": sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0,
            "scalar_busy_cycles_per_token": sq.get("SQ_ACTIVE_INST_SCA", 0.0) * 4.0,
            "valu_insts_per_token": sq.get("SQ_INSTS_VALU"), "scalar_insts_per_token": sq.get("SQ_INSTS_SALU"),
            "wave_cycles_per_token": sq.get("SQ_WAVE_CYCLES", 0.0) * 4.0,
            "kernel_ms_per_sweep_profiled": {"rdreq_pass": rd.get("_kernel_ms_sum_per_sweep"), "sq_pass": sq.get("_kernel_ms_sum_per_sweep"),
                                             "span_rdreq_pass": rd.get("_kernel_span_ms_per_sweep")},
            "sweeps_seen": {"rd": rd.get("_sweeps_seen"), "wr": wr.get("_sweeps_seen"), "sq": sq.get("_sweeps_seen")},
            "pmc_source": f"profiles/profile_r05.sh: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ / WRITE_SIZE passes ({tag}); a request = 128 B (profiles/r02_fetch_calibration.txt)",
            "sq_source": f"profiles/profile_r05.sh: rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA ... pass ({tag}); the counters tick once per four cycles"}


def one(W, tokens, warm, steps, views):
    w = W.lower()
    cmd = f"`bench.py --workload {W} --steps {steps} --warmup {warm} --no-cpu-baseline`"
    res = {"workload": W, "tokens_per_sweep": tokens}
    if views > 1:
        # the bench command's chain: warm + steps deferred sweeps, then 3 + 10 live, 3 + 10 segmented (8 segments), 3 + 10 deferred
        b = warm + steps
        res["deferred"] = block(w, tokens, cmd + ", the timed deferred sweeps", warm, steps, "draw_p_kernel")
        res["live"] = block(w, tokens, cmd + ", the ten timed MVHDP_SWEEP_LIVE sweeps", b + 3, 10, "draw_p_kernel")
        res["segmented"] = block(w, tokens, cmd + ", the ten timed MVHDP_SWEEP_SEGMENT_APPLY sweeps (8 segments)", b + 16, 10, "draw_p_kernel")
        res["deferred_same_age"] = block(w, tokens, cmd + ", the ten deferred sweeps behind them", b + 29, 10, "draw_p_kernel")
    else:
        res["deferred"] = block(w, tokens, cmd + " --live-steps 0, the timed deferred sweeps", warm, steps, "build_trees_kernel")
        if os.path.isdir(f"{OUT}/{w}live_rd"):
            res["live"] = block(w + "live", tokens, cmd + " --live --live-steps 0, the timed live sweeps", warm, steps, "live_coef_kernel")
    json.dump(res, open(f"{ROOT}/gpurun_out/r05_{w}_counters.json", "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if "source" not in kk}) for k, v in res.items()}, indent=1))


def merge():
    import glob
    out = {"workloads": {}}
    # the committed summaries first, then whatever this run profiled anew (a run that profiles one workload must not drop the others)
    for f in sorted(glob.glob(f"{ROOT}/profiles/r05_c?_counters.json")) + sorted(glob.glob(f"{ROOT}/gpurun_out/r05_c?_counters.json")):
        j = json.load(open(f))
        if not any(isinstance(v, dict) and v.get("fabric_read_requests_per_token") for v in j.values()):
            continue                                  # (a summary of passes that did not run)
        out["workloads"][j["workload"]] = {k: v for k, v in j.items() if isinstance(v, dict)}
    # the bare-gather ceilings of the access patterns: measured in round 4 on the same chip (tools/microbench/row_gather_ceiling.hip)
    try:
        r4 = json.load(open(f"{ROOT}/profiles/r04_roofline_inputs.json"))
        for k in ("gather_ceiling_GBs", "gather_ceiling_source", "gather_ceilings_GBs", "gather_ceilings_source", "gather_ceiling_workload"):
            if k in r4:
                out[k] = r4[k]
    except Exception:
        pass
    json.dump(out, open(f"{ROOT}/gpurun_out/r05_roofline_inputs.json", "w"), indent=1)


if sys.argv[2] == "merge":
    merge()
else:
    one(sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
