#!/bin/bash
# Round 3: vector / scalar / branch instructions per token of ALL sweep kernels of a 30-sweep chain from the random start, per workload
# (one rocprofv3 --pmc pass with --kernel-trace only; GPU box, through gpurun):   bash profiles/pmc_r03_insts_per_token.sh C4
R=$GRAFT_REPO_ROOT; W=$1
OUT=$R/gpurun_out/sqw_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT -- python3 $R/tools/per_sweep_times.py --workload $W --sweeps 30 > $OUT.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
agg = {}
disp = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sweep_fast_kernel<" in n:
        agg.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        disp[r["Dispatch_Id"]] = n.split("(")[0]
# total over the last 5 sweeps' kernels: take dispatches in order, group by sweep is hard; use totals over all sweep kernels / total tokens
from mvtopicmodel_amd import synth
c = synth.make_config("$W")
tok = c.total_tokens * 30
print("$W", {k: round(sum(v.values()) / tok, 2) for k, v in agg.items()}, "kernels", sorted(set(disp.values())))
PY
