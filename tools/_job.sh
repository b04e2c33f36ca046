python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_walk_threshold.py tests/test_gpu_kat7.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; tail -4 gpurun_out/r3_t8.log
python tools/per_sweep_times.py --sweeps 40 > gpurun_out/r3_ps_c4.json 2>/dev/null
for w in C2 C3 C5; do python tools/per_sweep_times.py --workload $w --sweeps 30 > gpurun_out/r3_ps_$w.json 2>/dev/null; done
python -c "
import json
for w in ['c4','C2','C3','C5']:
    d=json.load(open('gpurun_out/r3_ps_%s.json'%w)); k=d['kernel_ms']; print(w, round(sum(k[5:25])/20,3), k[-6:])
"
