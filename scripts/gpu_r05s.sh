# round 5 (s): four fp32 sums per lane (summation term of the bound ceil(B/4) + 11): A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
P=$PWD/kerneldensityestimate.jl_amd
( KDEHIP_LIB=$P/libkdehip_c4four.so NP=1100 timeout 120 python scripts/check_screen_chunk.py 3 8 5000 2048 || echo "FAILED c4"
  KDEHIP_LIB=$P/libkdehip_c3four.so NP=1100 timeout 120 python scripts/check_screen_chunk.py 6 4 2048 4096 8000 1000 || echo "FAILED c3" ) 2>&1 | grep -v amdgpu.ids | tee $O/check.txt
python scripts/ab_libs.py --libs $P/libkdehip_c4base.so $P/libkdehip_c4tight.so $P/libkdehip_c4four.so --configs c4 --rounds 5 --steps 5 2>&1 | tail -4 | tee $O/ab_c4.txt
python scripts/ab_libs.py --libs $P/libkdehip_c3base.so $P/libkdehip_c3tight.so $P/libkdehip_c3four.so --configs c3 --rounds 9 --steps 20 2>&1 | tail -4 | tee $O/ab_c3.txt
for l in c3tight c3four; do KDEHIP_LIB=$P/libkdehip_$l.so python scripts/chain_timing.py c3 10 2048 2>&1 | tail -1 | cut -c1-120; done | tee $O/chain.txt
