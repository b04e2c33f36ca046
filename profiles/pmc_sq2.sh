#!/bin/bash
# Second-level SQ counters of the C4 sweep kernel: where the wave cycles go (VALU / scalar / VMEM issue,
# instruction fetch, VMEM latency via SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD).
#   bash profiles/pmc_sq2.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r01}
OUT=$R/gpurun_out/sq2_$T
mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA}"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob, json
out = {}
for d in ("a", "b"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if "sweep_fast" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[2:] if len(v) > 2 else v
        out["%s | %s" % k] = sum(v) / len(v)
print(json.dumps(out, indent=1))
open("$OUT/summary.json", "w").write(json.dumps(out, indent=1))
PY
