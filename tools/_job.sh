python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_walk_threshold.py tests/test_gpu_kat7.py tests/test_gpu_live.py tests/test_gpu_segmented.py -m gpu -x -q > gpurun_out/r3_t11.log 2>&1; tail -4 gpurun_out/r3_t11.log
bash profiles/pmc_r03_sq.sh > gpurun_out/r3_sq2.log 2>&1; python -c "
import json
d=json.load(open('gpurun_out/r03_sq_summary.json')); print({k:round(v['per_token'],2) for k,v in d.items() if isinstance(v,dict)}); print(d['kernel_ms_last8'])
"
for w in C2 C3; do python tools/per_sweep_times.py --workload $w --sweeps 30 > gpurun_out/r3_ps3_$w.json 2>/dev/null; done
python tools/mode_times.py --mode deferred --sweeps 34 > gpurun_out/r3_m8_def.json 2>/dev/null
python -c "
import json
for w in ['C2','C3']:
    d=json.load(open('gpurun_out/r3_ps3_%s.json'%w)); k=d['kernel_ms']; print(w, round(sum(k[5:25])/20,3), k[-4:])
j=json.load(open('gpurun_out/r3_m8_def.json')); print('def', j['total_ms_sweeps_5_24'], j['total_ms_last_half'], j['kernel_ms'][:14])
"
