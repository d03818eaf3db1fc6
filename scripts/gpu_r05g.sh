# round 5 (g): where kde!(pGM) from device points spends its time after the search; host pool with polling workers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
nproc | tee $O/out.txt
for n in 0 3 7 15; do KDEHIP_HOST_THREADS=$n python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
import time, numpy as np, sys, os
sys.path.insert(0,'.')
import kdehip
rng=np.random.default_rng(0)
def T(f,n=50):
    f(); t=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t)/n*1e3
for N in (1000, 2048):
    pg=rng.standard_normal((6,N)); bw=np.full(6,0.3)
    print("host tree, worker threads", os.environ["KDEHIP_HOST_THREADS"], "N", N, round(T(lambda: kdehip.kde(pg,bw)),4), "ms")
PY
done
KDEHIP_LIB=$GRAFT_REPO_ROOT/kerneldensityestimate.jl_amd/libkdehip_r05e.so python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
import time, numpy as np, sys, os
sys.path.insert(0,'.')
import kdehip
rng=np.random.default_rng(0)
def T(f,n=50):
    f(); t=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t)/n*1e3
for N in (1000, 2048):
    pg=rng.standard_normal((6,N)); bw=np.full(6,0.3)
    print("host tree, round-4 pool (sleeping workers) N", N, round(T(lambda: kdehip.kde(pg,bw)),4), "ms")
PY
KDEHIP_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "from_device_points" | tail -5 | tee -a $O/out.txt
import sys; sys.path.insert(0, '.')
import numpy as np, torch, kdehip
P = torch.randn(6 * 2048, dtype=torch.float64, device="cuda:0")
for _ in range(6):
    x = kdehip.DeviceDensity.from_device_points(P, 6, 2048); x.close()
PY
python scripts/chain_timing.py c3 10 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
python scripts/chain_timing.py c3 10 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
