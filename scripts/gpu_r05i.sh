cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
L=kerneldensityestimate.jl_amd
python scripts/ab_libs.py --libs $L/libkdehip_nowords.so $L/libkdehip_words.so --configs c3 --rounds 11 --steps 20 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
python scripts/ab_libs.py --libs $L/libkdehip_nowords.so $L/libkdehip_words.so --configs c3 --nout 16384 --rounds 5 --steps 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
python scripts/ab_libs.py --libs $L/libkdehip_nowords.so $L/libkdehip_words.so --configs c3 --nout 256 --rounds 9 --steps 20 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
