#!/bin/bash
# Round-3 profiling recipe (GPU box, through gpurun):  bash profiles/profile_r03.sh
#  1. rocprofv3 --kernel-trace --stats of the driver's bench command (C4) -> kernel stats
#  2. the bench lines themselves: C4 (with the CPU leg), C5, C3, C2
#  3. memory-side PMC passes of the same command (separate --pmc passes with --kernel-trace only, MI355X_MICROARCH.md "HBM";
#     every TCC_EA0_RDREQ is a 128-byte request on gfx950, FETCH_SIZE tallies 64 B each: profiles/r02_fetch_calibration.txt)
#  4. LL curves of the live sweep on the 16-bit mirror against the CPU restatement of the reference (profiles/r02_ll_cpu.json)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT/c4
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --live-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $R/bench.py $ARGS > $OUT/c4.log 2>&1 || { tail -5 $OUT/c4.log; exit 1; }
cp $(ls -t $OUT/c4/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_c4_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B" "WRITE_SIZE" "TCC_HIT TCC_MISS"; do
  i=$((i+1)); mkdir -p $OUT/pmc/s$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc/s$i -- python3 $R/bench.py $ARGS > $OUT/pmc/s$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/pmc/s$i.log; exit 1; }
done
cd $R
PMC_TAG=r03 PMC_SOURCE=profiles/profile_r03.sh python3 profiles/pmc_r02b_summary.py $OUT/pmc > $OUT/pmc_summary.txt && cp $OUT/pmc/r03_c4_pmc_summary.json gpurun_out/
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_c4.json.log 2> $OUT/bench_c4.err || exit 1
timeout -k 10 400 python3 bench.py --workload C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_c5.json.log 2> $OUT/bench_c5.err || exit 1
timeout -k 10 300 python3 bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_c3.json.log 2> $OUT/bench_c3.err || exit 1
timeout -k 10 300 python3 bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_c2.json.log 2> $OUT/bench_c2.err || exit 1
timeout -k 10 300 python3 tools/ll_curves.py gpu --workload C3 --docs 200000 --sweeps 100 --every 5 --live-segments 4 --only live --live16 1 --out gpurun_out/r03_ll_gpu_live16.json > $OUT/ll.log 2>&1 || { tail -3 $OUT/ll.log; exit 1; }
python3 tools/ll_curves.py table profiles/r02_ll_cpu.json profiles/r02_ll_gpu.json gpurun_out/r03_ll_gpu_live16.json > gpurun_out/r03_ll_curves.md
rm -rf $OUT/c4 $OUT/pmc/s*
head -12 gpurun_out/r03_c4_kernel_stats.csv | cut -c1-170
for f in c4 c5 c3 c2; do tail -1 gpurun_out/r03_bench_$f.json.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e9,3), 'G tok/s', round(j['ms_per_step'],2), 'ms frac', round(j['roofline']['frac'],3), 'live', round(j.get('live',{}).get('value',0)/1e9,3), 'seg', round(j.get('segmented',{}).get('value',0)/1e9,3))"; done
cat gpurun_out/r03_c4_pmc_summary.json | head -30
tail -12 gpurun_out/r03_ll_curves.md
