# round 5 (d): ablation of the v2 screened step: requests ahead / hoisted header loads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_nokept.so $L/libkdehip_neither.so $L/libkdehip_noahead.so $L/libkdehip_nohoist.so $L/libkdehip_v2.so --configs c3 --rounds 9 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
