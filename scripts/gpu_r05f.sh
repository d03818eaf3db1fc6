# round 5 (f): packed step words (no descriptor load per step) + kde!(pGM) with the host work under the search
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_r05e.so $L/libkdehip.so --configs c3 c2 --rounds 9 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
python scripts/ab_libs.py --libs $L/libkdehip_r05e.so $L/libkdehip.so --configs c4 --rounds 5 --steps 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
python scripts/ab_libs.py --libs $L/libkdehip_r05e.so $L/libkdehip.so --configs c3 --nout 16384 --rounds 5 --steps 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_screen.py tests/test_gpu_lean.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_device_density.py tests/test_gpu_batch.py tests/test_gpu_bandwidth.py -q -m gpu -x > $O/tests.txt 2>&1; tail -n 5 $O/tests.txt
python scripts/chain_timing.py c3 10 2>&1 | grep -v amdgpu.ids | tee $O/chain.txt
python scripts/chain_timing.py c3 10 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/chain.txt
KDEHIP_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "from_device_points\|auto_bandwidth" | tail -6 | tee -a $O/chain.txt
import sys; sys.path.insert(0, '.')
import numpy as np, torch, kdehip
P = torch.randn(6 * 2048, dtype=torch.float64, device="cuda:0")
for _ in range(4):
    x = kdehip.DeviceDensity.from_device_points(P, 6, 2048); x.close()
PY
