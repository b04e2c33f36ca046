#!/bin/bash
# usage: pmc_run.sh tag  (runs one SQ pass at 200k docs)
R=$GRAFT_REPO_ROOT; T=$1
mkdir -p $R/gpurun_out/pmc_$T && cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_$T -- python3 $R/bench.py --docs 200000 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$T.log 2>&1
cd $R && python3 profiles/pmc_summary.py gpurun_out/pmc_$T 29447726
