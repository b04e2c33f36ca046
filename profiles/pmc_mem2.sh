#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mem2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline"
i=0
for set in "TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES" \
           "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM TCC_TAG_STALL" \
           "TCC_EA0_ATOMIC_LEVEL TCC_EA0_ATOMIC TCC_BUSY TCC_CYCLE" \
           "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TD_TCP_STALL_CYCLES TCP_TCP_LATENCY" \
           "TCC_REQ TCC_STREAMING_REQ TCC_NC_REQ TCC_RW_REQ"; do
  i=$((i+1)); mkdir -p $OUT/p$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
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
    print("%-40s %18.0f  per token %10.3f" % (k, out[k], out[k] / tok))
PY
