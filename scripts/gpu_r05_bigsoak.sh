# round 5: long soak of the final binary (GPU box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bigsoak; mkdir -p $O
( timeout 1500 python scripts/soak_screen.py 1500 2>&1 | tail -3
  KDEHIP_FUZZ_N=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
  timeout 900 python scripts/soak_threads.py 16 4000 2>&1 | tail -1
  timeout 900 python scripts/soak_multi.py 6000 --resident 2>&1 | tail -1
  timeout 1500 python scripts/soak_chunked.py 2000 2>&1 | tail -1
  timeout 900 python scripts/soak_fp32.py 3000 2>&1 | tail -1
  timeout 900 python scripts/soak_determinism.py 3000 6 2>&1 | tail -1
  timeout 1500 python scripts/soak_callers.py 3000 2>&1 | tail -1 ) > $O/soaks.txt 2>&1
cat $O/soaks.txt
