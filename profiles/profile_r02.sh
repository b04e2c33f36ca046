#!/bin/bash
# Round-2 profiling recipe (GPU box, through gpurun):  bash profiles/profile_r02.sh [tag]      (tag r02b: the build with the thresholded tree walk)
# kernel trace + stats of the default bench command (C4) and of the C5 bench; the bench lines themselves (with the CPU leg
# for C4); PMC passes are profiles/calib.sh.
T=${1:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$T
mkdir -p $OUT/c4 $OUT/c5 $OUT/c4live
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --live-steps 0 > $OUT/c4.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 $R/bench.py --workload C5 --steps 10 --warmup 3 --no-cpu-baseline --live-steps 0 > $OUT/c5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4live -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --live > $OUT/c4live.log 2>&1 || exit 1
cd $R
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_c4.json.log 2>&1 || exit 1
timeout -k 10 400 python3 bench.py --workload C5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3.json.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2.json.log 2>&1 || exit 1
for f in c4 c5 c4live; do echo "== $f"; head -8 $OUT/$f/*/*kernel_stats.csv | cut -c1-160; done
for f in c4 c5 c3 c2; do tail -1 $OUT/bench_$f.json.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$f', round(j['value']/1e9,3), 'G tok/s', round(j['ms_per_step'],2), 'ms frac', round(j['roofline']['frac'],3), 'live', j.get('live',{}).get('value'))"; done
