#!/bin/bash
# Memory-side counters of the sweep kernels of the DRIVER's bench command (C4, --steps 20 --warmup 5) for the build with the
# thresholded tree walk (the walk threshold moves during these sweeps, so the bytes per token are those of this window, not a
# constant of the build).  Separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md); calibration of the counters:
# profiles/calib.sh / r02_fetch_calibration.txt (every TCC_EA0_RDREQ is a 128-byte request; FETCH_SIZE tallies 64 B each).
#   bash profiles/pmc_r02b.sh            (on the GPU box, through gpurun)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_r02b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --live-steps 0"
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" "WRITE_SIZE" "TCC_HIT TCC_MISS"; do
  i=$((i+1)); mkdir -p $OUT/s$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/bench.py $ARGS > $OUT/s$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/s$i.log; exit 1; }
done
cd $R
python3 profiles/pmc_r02b_summary.py $OUT | tee $OUT/summary.txt
