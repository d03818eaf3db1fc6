# round 5 (h): kde!(pGM) from device points with the pinned mirror + cheap flags; what np.empty costs beside a live HIP runtime
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_device_density.py tests/test_gpu_bandwidth.py tests/test_gpu_treebuild.py -q -m gpu -x 2>&1 | tail -n 3 | tee $O/out.txt
KDEHIP_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "from_device_points" | tail -4 | tee -a $O/out.txt
import sys; sys.path.insert(0, '.')
import numpy as np, torch, kdehip
P = torch.randn(6 * 2048, dtype=torch.float64, device="cuda:0")
for _ in range(6):
    x = kdehip.DeviceDensity.from_device_points(P, 6, 2048); x.close()
PY
python scripts/chain_timing.py c3 10 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
python scripts/chain_timing.py c3 10 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
import time, numpy as np
def T(f, n=200):
    f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e6
def alloc():
    a = np.empty(150000); a[::512] = 1.0; del a
print("np.empty(1.2 MB) + touch + free, no GPU runtime in the process: %.0f us" % T(alloc))
import torch
torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
print("the same with a live HIP runtime: %.0f us" % T(alloc))
PY
