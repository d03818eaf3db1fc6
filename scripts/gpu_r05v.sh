# round 5 (v): after the speculative LOOCV rounds: the suite, the soaks that search bandwidths, the timings that contain a search
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
( timeout 600 python scripts/soak_threads.py 16 1500 2>&1 | tail -1
  timeout 900 python scripts/soak_callers.py 1500 2>&1 | tail -1 ) | tee $O/soaks.txt
python scripts/chain_timing.py c3 10 2>&1 | tail -2 | tee $O/chain.txt
python scripts/chain_timing.py c3 10 2048 2>&1 | tail -2 | tee -a $O/chain.txt
python scripts/pipeline_timing.py c3 2>&1 | grep -v amdgpu | tee $O/pipeline.txt
python scripts/loocv_timing.py 20 2>&1 | tail -1 | tee $O/loocv.txt
python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_form.json 2> $O/bench.err; cut -c1-400 $O/bench_c3_driver_form.json
