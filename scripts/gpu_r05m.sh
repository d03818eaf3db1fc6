# round 5 (m): streamed fp32 screening (screen tiles one per step through the pool halves): suite + A/B against
# KDEHIP_SCREEN_STREAM=0 (round 5's resident-only screening) on the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_screen.py tests/test_gpu_lean.py tests/test_gpu_parity.py -x -q 2>&1 | tail -6 | tee $O/tests.log
for r in 1 2; do
for s in 1 0; do
  for c in c4 c3; do
    KDEHIP_SCREEN_STREAM=$s python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('stream=$s', '$c', 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'], 'screen', d.get('screen'))"
  done
  KDEHIP_SCREEN_STREAM=$s python scripts/chain_timing.py c3 10 2048 2>&1 | tail -3
done; done | tee $O/ab.txt
KDEHIP_FUZZ_N=600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2 | tee $O/fuzz.txt
timeout 600 python scripts/soak_chunked.py 400 2>&1 | tail -1 | tee -a $O/fuzz.txt
timeout 600 python scripts/soak_determinism.py 600 4 2>&1 | tail -1 | tee -a $O/fuzz.txt
