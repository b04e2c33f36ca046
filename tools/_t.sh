cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_walk_threshold.py tests/test_gpu_parity.py tests/test_gpu_segmented.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
MVHDP_FORCE_RMAX=1 MVHDP_WALK_THETA=0.7,0.7,0.7,0.7,0.7,0.7,0.7,0.7 MVHDP_FUZZ_CASES=120 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
for nv in 0 1; do
MVHDP_NARROW=$nv timeout -k 10 300 python tools/per_sweep_times.py --workload C4 --sweeps 50 2>/dev/null | python -c "
import sys,json; k=json.loads(sys.stdin.read())['kernel_ms']; print('narrow=$nv C4 5-24 %.3f 25-39 %.3f 40-49 %.3f'%(sum(k[5:25])/20,sum(k[25:40])/15,sum(k[40:])/10))"
done
