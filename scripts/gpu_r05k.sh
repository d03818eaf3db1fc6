cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_base.so $L/libkdehip_lvar.so --configs c3 --rounds 11 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
