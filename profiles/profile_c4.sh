#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):
#   bash profiles/profile_c4.sh <tag>
# 1) kernel trace + stats of the default bench command, 2) FETCH_SIZE pass, 3) WRITE_SIZE pass
# (PMC passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
R=$GRAFT_REPO_ROOT; T=${1:-r01}
OUT=$R/gpurun_out/prof_$T
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json, collections
out = {}
tok = 147225025
# the dominant sweep kernel of the run (largest total time in the --stats pass)
stats = list(csv.DictReader(open(glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0])))
dom = max((r for r in stats if "sweep" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))["Name"]
out["dominant_kernel"] = dom
for d in ("fetch", "write", "sq"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == dom:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[2:] if len(v) > 2 else v          # drop the warm-up sweeps
        out[k] = sum(v) / len(v)
out["kernel_stats_csv"] = open(glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]).read()
out["tokens_per_launch"] = tok
print(json.dumps({k: v for k, v in out.items() if k != "kernel_stats_csv"}, indent=1))
open("$OUT/summary.json", "w").write(json.dumps(out, indent=1))
PY
tail -1 $OUT/stats.log | cut -c1-300
