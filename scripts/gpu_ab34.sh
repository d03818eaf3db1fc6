# interleaved A/B of development libraries for c3 and c4: TAGS="base lin" bash scripts/gpu_ab34.sh
cd $GRAFT_REPO_ROOT
P=$PWD/kerneldensityestimate.jl_amd
L3=""; L4=""
for t in $TAGS; do L3="$L3 $P/libkdehip_c3$t.so"; L4="$L4 $P/libkdehip_c4$t.so"; done
python scripts/ab_libs.py --libs $L3 --configs c3 --rounds 9 --steps 20 2>&1 | tail -$(( $(echo $TAGS | wc -w) + 1 ))
python scripts/ab_libs.py --libs $L4 --configs c4 --rounds 5 --steps 5 2>&1 | tail -$(( $(echo $TAGS | wc -w) + 1 ))
for t in $TAGS; do KDEHIP_LIB=$P/libkdehip_c3$t.so python scripts/chain_timing.py c3 6 2048 2>&1 | tail -1 | cut -c1-100; done
