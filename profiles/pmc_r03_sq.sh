#!/bin/bash
# Round 3: what is the settled C4 sweep kernel (1-round variant on the 16-bit mirror) bound by?  SQ counters of the last sweeps of a
# 36-sweep chain (tools/per_sweep_times.py), one rocprofv3 pass per counter group (--pmc with --kernel-trace only).
#   bash profiles/pmc_r03_sq.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_r03
mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $R/tools/per_sweep_times.py --sweeps 36 > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/b -- python3 $R/tools/per_sweep_times.py --sweeps 36 > $OUT/b.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json
TOK = 147225025.0
out = {}
for d in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sweep_fast_kernel<1, false, true, true" in n:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[-8:]
        out[k] = {"mean_per_launch": sum(v) / len(v), "per_token": sum(v) / len(v) / TOK}
kt = glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt)) if "sweep_fast_kernel<1, false, true, true" in r["Kernel_Name"]]
out["kernel_ms_last8"] = d[-8:]
print(json.dumps(out, indent=1))
open("$R/gpurun_out/r03_sq_summary.json", "w").write(json.dumps(out, indent=1))
PY
rm -rf $OUT/a $OUT/b
