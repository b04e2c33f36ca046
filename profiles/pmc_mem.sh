#!/bin/bash
# Memory-side counters of the C4 sweep kernel (separate passes, --kernel-trace only):
# L1 (TCP) accesses / misses / translation, L2 (TCC) hits / misses / atomics / fabric requests.
#   bash profiles/pmc_mem.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r01}
OUT=$R/gpurun_out/mem_$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA}"
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" \
           "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_MULTI_MISS" \
           "TCC_HIT TCC_MISS TCC_READ TCC_ATOMIC" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_ATOMIC" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_TCC_ATOMIC_WITHOUT_RET_REQ"; do
  i=$((i+1)); mkdir -p $OUT/p$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections, json
tok = 147225025
out = {}
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sweep_fast_kernel<2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[2:] if len(v) > 2 else v
        out[k] = sum(v) / len(v)
for k in sorted(out):
    print("%-40s %16.0f  per token %10.3f" % (k, out[k], out[k] / tok))
open("$OUT/summary.json", "w").write(json.dumps(out, indent=1))
PY
