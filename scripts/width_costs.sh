#!/bin/bash
# One full round of workgroups (256 = one per CU) at every workgroup width: the relative costs behind
# chains_per_workgroup() in csrc/gibbs_dispatch.cpp.  Runs on the GPU box: gpurun -- bash scripts/width_costs.sh
# End of round 2 (lean kernel), config 3: 0.480 / 0.653 / 0.920 / 1.135 ms for 4 / 8 / 12 / 16 chains per workgroup
# (= 1 : 1.36 : 1.92 : 2.36; the policy's constants are 1 : 1.34 : 1.86 : 2.31); config 2: 1 : 1.20 : 1.57 : 1.91;
# config 4: 1 : 1.26 : 1.83 (general kernel) : 2.06.
for cfg in c3 c2 c4; do
for wv in "2 1024" "8 2048" "12 3072" "16 4096"; do set -- $wv
python scripts/ab_libs.py --libs kerneldensityestimate.jl_amd/libkdehip.so --configs $cfg --rounds 3 --steps 8 --variant $1 --nout $2 2>&1 | grep "Nout=" | sed "s/^/width $1: /"
done; done
