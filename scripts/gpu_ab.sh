# scripts/gpu_ab.sh "<lib tags>" "<configs>" [extra ab_libs args]: interleaved A/B of development libraries on the GPU box
cd $GRAFT_REPO_ROOT
L=kerneldensityestimate.jl_amd
LIBS=""; for t in $1; do LIBS="$LIBS $L/libkdehip_$t.so"; done
python scripts/ab_libs.py --libs $LIBS --configs $2 --rounds 9 --steps 20 $3 2>&1 | grep -v amdgpu.ids
