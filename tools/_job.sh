cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 25 0; do
MVHDP_FORK_DELAY_US=$d python tools/mode_times.py --mode seg8 --sweeps 34 > gpurun_out/r3_m6_seg8_d$d.json 2>/dev/null
MVHDP_FORK_DELAY_US=$d python tools/mode_times.py --mode live4 --sweeps 34 > gpurun_out/r3_m6_live4_d$d.json 2>/dev/null
MVHDP_FORK_DELAY_US=$d python tools/mode_times.py --mode deferred --sweeps 34 > gpurun_out/r3_m6_def_d$d.json 2>/dev/null
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_seg8 -o trace -- python3 tools/mode_times.py --mode seg8 --sweeps 28 > /dev/null 2> gpurun_out/r3_p2_seg8.err
python tools/kernel_timeline.py gpurun_out/prof_seg8 --last 1 > gpurun_out/r3_tl2_seg8.txt
rm -rf gpurun_out/prof_seg8
python -c "
import json
for m in ['seg8','live4','def']:
  for d in [25,0]:
    j=json.load(open('gpurun_out/r3_m6_%s_d%d.json'%(m,d))); print(m,d, j['total_ms_sweeps_5_24'], j['total_ms_last_half'])
"
head -30 gpurun_out/r3_tl2_seg8.txt
