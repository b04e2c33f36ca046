#!/bin/bash
# Round-5 profiling recipe (GPU box, through gpurun):  bash profiles/profile_r05.sh [C4|C3|C2|C5 ...]   (default: C4 C3)
# For every workload W, over the WHOLE chain its bench line times -- the deferred window, then the live, the segmented and the same-age
# deferred sweeps (bench.py --workload W --steps S --warmup Wm) --:
#   1. rocprofv3 --kernel-trace --stats                                    -> gpurun_out/r05_<w>_kernel_stats.csv
#   2. memory-side PMC passes, one counter group per run (MI355X_MICROARCH.md "HBM"): TCC_EA0_RDREQ (every request is 128 bytes on
#      gfx950: profiles/r02_fetch_calibration.txt) and WRITE_SIZE
#   3. an SQ pass: cycles a SIMD's vector ALU / scalar unit is busy, instructions, wave cycles
# cut into the windows of the modes by the sweeps' marker kernel (profiles/r05_counters.py).  Summaries: gpurun_out/r05_<w>_counters.json,
# gpurun_out/r05_roofline_inputs.json (copied to profiles/ by hand: bench.py reads it from there).
R=$GRAFT_REPO_ROOT
if [ -z "$R" ]; then echo "GRAFT_REPO_ROOT is not set (run through gpurun)"; exit 1; fi
OUT=$R/gpurun_out/prof_r05
WL=${@:-C4 C3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <dir> <counters or ""> -- <program ...>
  local d=$1 c=$2; shift 3
  mkdir -p $d
  if [ -z "$c" ]; then timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@" > $d.log 2>&1
  else timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- "$@" > $d.log 2>&1; fi
  local rc=$?; if [ $rc -ne 0 ]; then echo "profile run failed ($d)"; tail -3 $d.log; exit 1; fi
}
for W in $WL; do
  w=$(echo $W | tr A-Z a-z)
  case $W in C5) S=10; Wm=3;; *) S=20; Wm=5;; esac
  VIEWS=$(cd $R && python3 -c "from mvtopicmodel_amd import synth; print(len(synth.CONFIGS['$W']['V']))")
  ARGS="--workload $W --steps $S --warmup $Wm --no-cpu-baseline"
  if [ $VIEWS = 1 ]; then ARGS="$ARGS --live-steps 0"; fi
  TOK=$(cd $R && python3 -c "from mvtopicmodel_amd import synth; print(int(synth.config_doc_token_counts('$W').sum()))")
  prof $OUT/${w}_stats "" -- python3 $R/bench.py $ARGS
  cp $(ls -t $OUT/${w}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_${w}_kernel_stats.csv
  echo "$W kernel stats done"
  prof $OUT/${w}_rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" -- python3 $R/bench.py $ARGS
  echo "$W rd pass done"
  prof $OUT/${w}_wr "WRITE_SIZE" -- python3 $R/bench.py $ARGS
  echo "$W wr pass done"
  prof $OUT/${w}_sq "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 $R/bench.py $ARGS
  echo "$W sq pass done"
  if [ $VIEWS = 1 ]; then
    # a single-view model has no per-sweep marker that every mode shares: its live sweeps are profiled as a command of their own
    # (bench.py --live: every sweep a live one; marker = live_coef_kernel, once per live-rows sweep)
    LARGS="--workload $W --steps $S --warmup $Wm --no-cpu-baseline --live --live-steps 0"
    prof $OUT/${w}live_rd "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" -- python3 $R/bench.py $LARGS
    prof $OUT/${w}live_wr "WRITE_SIZE" -- python3 $R/bench.py $LARGS
    prof $OUT/${w}live_sq "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 $R/bench.py $LARGS
    echo "$W live passes done"
  fi
  (cd $R && python3 profiles/r05_summarize.py $OUT $W $TOK $Wm $S $VIEWS) || exit 1
done
cd $R && python3 profiles/r05_summarize.py $OUT merge
cat gpurun_out/r05_roofline_inputs.json | head -c 3000
