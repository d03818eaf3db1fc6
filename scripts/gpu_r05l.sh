# round 5 (l): per-level instruction / wait / time table of config 5 (fp32, 16 chains per workgroup); cold start pieces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
L=$PWD/kerneldensityestimate.jl_amd/libkdehip_exp5.so
KDEHIP_LEVEL_CONFIG=c5 KDEHIP_LEVELS=14 KDEHIP_CHAINS=8192 KDEHIP_STEPS_PER_LEVEL=84 bash scripts/level_profile.sh $L > $O/level_insts_c5.txt 2> $O/level_insts_c5.err
KDEHIP_LIB=$L python scripts/level_timing2.py c5 0 2>&1 | grep -v amdgpu.ids >> $O/level_insts_c5.txt
cat $O/level_insts_c5.txt
for i in 1 2 3; do python scripts/cold_pieces.py 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/cold.txt
