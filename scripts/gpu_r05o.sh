# round 5 (o): chunked + streamed screening in the full library: suite, soaks, bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
P=$PWD/kerneldensityestimate.jl_amd
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/tests.log
( NP=1100 timeout 300 python scripts/check_screen_chunk.py 6 4 2048 4096 8000 3000
  NP=4096 timeout 300 python scripts/check_screen_chunk.py 3 8 5000 10000 2048
  NP=1100 timeout 300 python scripts/check_screen_chunk.py 2 3 20000 9000
  NP=600 timeout 300 python scripts/check_screen_chunk.py 8 2 6000 ) 2>&1 | grep -v amdgpu.ids | tee $O/check.txt
if [ -f $P/libkdehip_base.so ]; then
python scripts/ab_libs.py --libs $P/libkdehip_base.so $P/libkdehip.so --configs c3 --rounds 7 --steps 20 2>&1 | tail -3 | tee $O/ab.txt
python scripts/ab_libs.py --libs $P/libkdehip_base.so $P/libkdehip.so --configs c4 --rounds 5 --steps 5 2>&1 | tail -3 | tee -a $O/ab.txt
fi
for c in c3 c4; do python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$c', 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'], 'screen', d.get('screen'))"; done | tee $O/bench.txt
python bench.py --config c4 --strong --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $O/bench_c4_strong.json | cut -c1-300
python scripts/chain_timing.py c3 10 2048 2>&1 | tail -2 | tee $O/chain.txt
python scripts/chain_timing.py c3 10 2>&1 | tail -2 | tee -a $O/chain.txt
( KDEHIP_FUZZ_N=1000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
  timeout 600 python scripts/soak_chunked.py 600 2>&1 | tail -1
  timeout 600 python scripts/soak_determinism.py 1000 4 2>&1 | tail -1
  timeout 600 python scripts/soak_multi.py 1500 --resident 2>&1 | tail -1
  timeout 600 python scripts/soak_callers.py 600 2>&1 | tail -1 ) 2>&1 | tee $O/soaks.txt
