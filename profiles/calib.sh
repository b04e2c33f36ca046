#!/bin/bash
# Calibrate the memory-side read counters on access patterns whose touched bytes are known by construction
# (tools/microbench/fetch_calib.hip), then collect the same counters on the C4 sweep kernel.
#   bash profiles/calib.sh <tag>          (on the GPU box, through gpurun)
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md).
R=$GRAFT_REPO_ROOT; T=${1:-r02}
OUT=$R/gpurun_out/calib_$T
mkdir -p $OUT
BIN=$R/tools/microbench/fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $R/tools/microbench/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
$BIN > $OUT/truth.jsonl || exit 1
i=0
for set in "FETCH_SIZE" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" \
           "TCC_HIT TCC_MISS TCC_READ TCC_REQ" \
           "TCC_BUBBLE TCC_EA0_RDREQ_DRAM TCP_TCC_READ_REQ"; do
  i=$((i+1)); mkdir -p $OUT/m$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/m$i -- $BIN > $OUT/m$i.log 2>&1 || { echo "microbench pass $i ($set) failed"; tail -3 $OUT/m$i.log; }
done
# the same counters on the sweep kernel (C4, default bench command without the CPU leg)
ARGS="--steps 5 --warmup 2 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" \
           "TCC_HIT TCC_MISS TCC_READ TCC_REQ" \
           "TCC_BUBBLE TCC_EA0_RDREQ_DRAM TCP_TCC_READ_REQ" \
           "WRITE_SIZE"; do
  i=$((i+1)); mkdir -p $OUT/s$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/bench.py $ARGS > $OUT/s$i.log 2>&1 || { echo "sweep pass $i ($set) failed"; tail -3 $OUT/s$i.log; }
done
cd $R
python3 profiles/calib_summary.py $OUT | tee $OUT/summary.txt
