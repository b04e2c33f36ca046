#!/bin/bash
# Kernel trace of the C5 (K=1000, 5 views, 200k entities on one GPU) bench: how the sweep time splits
# between the primary register-resident pass, the 16-round pass and the generic LDS pass.
#   bash profiles/profile_c5.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r01}
OUT=$R/gpurun_out/prof_c5_$T
mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
ARGS="--workload C5 --docs 200000 --steps 4 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1 || exit 1
cd $R
f=$(ls $OUT/stats/*/*_kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
