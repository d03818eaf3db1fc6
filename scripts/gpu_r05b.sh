# round 5 (b): screen second pass from kept values vs re-evaluation (A/B), per-level times of the screened kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_nokept.so $L/libkdehip_kept.so --configs c3 --rounds 9 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
python scripts/ab_libs.py --libs $L/libkdehip_nokept.so $L/libkdehip_kept.so --configs c3 --rounds 5 --steps 20 --variant 5 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
KDEHIP_LIB=$GRAFT_REPO_ROOT/$L/libkdehip_exp.so python scripts/level_timing2.py c3 0 2>&1 | grep -v amdgpu.ids | tee $O/levels.txt
