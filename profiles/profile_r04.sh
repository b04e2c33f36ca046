#!/bin/bash
# Round-4 profiling recipe (GPU box, through gpurun):  bash profiles/profile_r04.sh [C4|C3|C2|C5 ...]   (default: all four)
# For every workload W, over the window its bench line times (bench.py --workload W --steps S --warmup Wm, deferred sweeps only):
#   1. rocprofv3 --kernel-trace --stats                                    -> gpurun_out/r04_<w>_kernel_stats.csv
#   2. memory-side PMC passes, one counter group per run (MI355X_MICROARCH.md "HBM"): TCC_EA0_RDREQ (every request is 128 bytes on
#      gfx950: profiles/r02_fetch_calibration.txt), WRITE_SIZE, and for C4 FETCH_SIZE, TCC_HIT/TCC_MISS
#   3. an SQ pass: cycles a SIMD's vector ALU / scalar unit is busy, instructions, wave cycles
# and for C4 the same two kinds of passes over sweeps 40-49 of a 50-sweep chain (the settled kernel), plus the bare-gather ceiling of
# the access pattern (tools/microbench/row_gather_ceiling.hip).  The summaries land in gpurun_out/r04_roofline_inputs.json (copied
# to profiles/ by hand: bench.py reads it from there) and gpurun_out/r04_<w>_counters.json.
R=$GRAFT_REPO_ROOT
if [ -z "$R" ]; then echo "GRAFT_REPO_ROOT is not set (run through gpurun)"; exit 1; fi
OUT=$R/gpurun_out/prof_r04
WL=${@:-C4 C3 C2 C5}
mkdir -p $OUT
cd $R/tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o row_gather_ceiling row_gather_ceiling.hip 2>/dev/null
cd $R && timeout -k 10 120 tools/microbench/row_gather_ceiling > gpurun_out/r04_row_gather_ceiling.txt 2>&1; cat gpurun_out/r04_row_gather_ceiling.txt
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <dir> <counters or ""> -- <program ...>
  local d=$1 c=$2; shift 3
  mkdir -p $d
  if [ -z "$c" ]; then timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@" > $d.log 2>&1
  else timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- "$@" > $d.log 2>&1; fi
  local rc=$?; if [ $rc -ne 0 ]; then echo "profile run failed ($d)"; tail -3 $d.log; exit 1; fi
}
for W in $WL; do
  w=$(echo $W | tr A-Z a-z)
  case $W in C5) S=10; Wm=3;; *) S=20; Wm=5;; esac
  ARGS="--workload $W --steps $S --warmup $Wm --no-cpu-baseline --live-steps 0"
  TOK=$(cd $R && python3 -c "from mvtopicmodel_amd import synth; print(int(synth.config_doc_token_counts('$W').sum()))")
  prof $OUT/${w}_stats "" -- python3 $R/bench.py $ARGS
  cp $(ls -t $OUT/${w}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/r04_${w}_kernel_stats.csv
  prof $OUT/${w}_rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" -- python3 $R/bench.py $ARGS
  prof $OUT/${w}_wr "WRITE_SIZE" -- python3 $R/bench.py $ARGS
  prof $OUT/${w}_sq "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 $R/bench.py $ARGS
  if [ $W = C4 ]; then
    prof $OUT/${w}_fs "FETCH_SIZE" -- python3 $R/bench.py $ARGS
    prof $OUT/${w}_hit "TCC_HIT TCC_MISS" -- python3 $R/bench.py $ARGS
    prof $OUT/${w}_set_rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" -- python3 $R/tools/per_sweep_times.py --workload C4 --sweeps 50
    prof $OUT/${w}_set_wr "WRITE_SIZE" -- python3 $R/tools/per_sweep_times.py --workload C4 --sweeps 50
    prof $OUT/${w}_set_sq "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 $R/tools/per_sweep_times.py --workload C4 --sweeps 50
  fi
  (cd $R && python3 profiles/r04_summarize.py $OUT $W $TOK $Wm $S) || exit 1
done
cd $R && python3 profiles/r04_summarize.py $OUT merge
cat gpurun_out/r04_roofline_inputs.json
