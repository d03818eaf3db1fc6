# round 5 (n): chunked fp32 screening: correctness, then A/B of development libraries
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
P=$PWD/kerneldensityestimate.jl_amd
( KDEHIP_LIB=$P/libkdehip_c3n.so timeout 300 python scripts/check_screen_chunk.py 6 4 2048 4096 8000 3000
  KDEHIP_LIB=$P/libkdehip_c4n.so timeout 300 python scripts/check_screen_chunk.py 3 8 5000 10000 2048 ) 2>&1 | grep -v amdgpu.ids | tee $O/check.txt
python scripts/ab_libs.py --libs $P/libkdehip_base.so $P/libkdehip_c3n.so --configs c3 --rounds 7 --steps 20 2>&1 | tail -4 | tee $O/ab_c3.txt
python scripts/ab_libs.py --libs $P/libkdehip_base.so $P/libkdehip_c4n.so --configs c4 --rounds 5 --steps 5 2>&1 | tail -4 | tee $O/ab_c4.txt
KDEHIP_LIB=$P/libkdehip_c3n.so python scripts/chain_timing.py c3 10 2048 2>&1 | tail -2 | tee $O/chain.txt
