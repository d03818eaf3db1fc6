#!/bin/bash
# Diagnostic build with in-kernel phase stamps (s_memtime); runs on the GPU box.  Never quote its run time.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO/kerneldensityestimate.jl_amd/csrc
mkdir -p $REPO/gpurun_out/stamps
for f in balltree.cpp pack_levels.cpp gibbs_dispatch.cpp devmem.cpp; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -x hip --offload-arch=gfx950 -c $f -o $REPO/gpurun_out/stamps/$f.o & done
for f in product.hip evaluate.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -c $f -o $REPO/gpurun_out/stamps/$f.o & done
for d in 1 2 3 4 5 6 7 8; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DKDEHIP_STAMPS -DKDEHIP_EXPERIMENTS -mllvm -disable-vector-combine -DKDEHIP_DIM=$d -c gibbs_kernel.hip -o $REPO/gpurun_out/stamps/gibbs_kernel_d$d.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $REPO/gpurun_out/stamps/libkdehip_stamps.so $REPO/gpurun_out/stamps/*.o
cd $REPO
KDEHIP_LIB=$REPO/gpurun_out/stamps/libkdehip_stamps.so python scripts/stamps.py
if [ "$1" = "all" ]; then
  KDEHIP_LIB=$REPO/gpurun_out/stamps/libkdehip_stamps.so python scripts/level_timing.py
  KDEHIP_LIB=$REPO/gpurun_out/stamps/libkdehip_stamps.so python scripts/ablate.py
fi
rm -f $REPO/gpurun_out/stamps/*.o $REPO/gpurun_out/stamps/*.so
