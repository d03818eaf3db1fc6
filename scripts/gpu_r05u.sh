# round 5 (u): speculative LOOCV rounds: tests first, then timings with and without
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bandwidth.py tests/test_gpu_chain.py tests/test_gpu_device_density.py -x -q 2>&1 | tail -5 | tee $O/tests.log
for s in 1 0; do for D in 6 3 1; do KDEHIP_LOOCV_SPEC=$s timeout 300 python scripts/loocv_sizes.py $D 2>&1 | grep -v amdgpu | sed "s/^/spec=$s /"; done; done | tee $O/sizes.txt
for s in 1 0; do KDEHIP_LOOCV_SPEC=$s python scripts/chain_timing.py c3 10 2>&1 | tail -2 | sed "s/^/spec=$s /"; KDEHIP_LOOCV_SPEC=$s python scripts/chain_timing.py c3 10 2048 2>&1 | tail -2 | sed "s/^/spec=$s /"; done | tee $O/chain.txt
timeout 600 python scripts/soak_callers.py 600 2>&1 | tail -1 | tee $O/soak.txt
