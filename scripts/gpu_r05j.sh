# round 5 (j): the persistent LOOCV search -- tests, timing against the launch-per-round search
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bandwidth.py tests/test_gpu_chain.py tests/test_gpu_screen.py -q -m gpu -x 2>&1 | tail -n 5 | tee $O/out.txt
for p in 1 0; do KDEHIP_LOOCV_PERSISTENT=$p KDEHIP_TIMING=1 python scripts/loocv_timing.py 20 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/out.txt; done
python scripts/chain_timing.py c3 10 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
python scripts/chain_timing.py c3 10 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
